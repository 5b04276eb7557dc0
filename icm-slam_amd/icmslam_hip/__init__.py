"""MI355X-native offline ICM sweep: ctypes binding (`_lib`) and host driver (`engine`)."""
from .engine import IcmError, SweepEngine, bearing_tables, cluster_first_scan, filtrar_map, prefilter_scans  # noqa: F401
