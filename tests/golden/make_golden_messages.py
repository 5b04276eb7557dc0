#!/usr/bin/env python3
"""Fixture generator for the I/O edge: the ROS messages the REFERENCE's publisher emits.

Runs only in the build container (the reference checkout at /root/reference never travels).  Imports the reference's
scripts/matlab2ros/createbag.py with an empty stub for the missing `roslibpy` dependency (its `__main__` block, the
only part that touches a ROS network, does not run on import) and calls its pure functions -- `mat2laser_scann`,
`mat2odometry`, `Header.new_message` (createbag.py:37-121) -- on scripts/data_IJAC2018.mat exactly as its publishing
loop does (createbag.py:124-147): one LaserScan and one Odometry message per sample, sequence numbers from two
independent `Header` objects.  Stores the message dicts of samples 0, 1, 100 and 1832 as JSON
(tests/golden/createbag_messages.json).  Data only: no reference source text is stored.

    python tests/golden/make_golden_messages.py
"""
import json
import os
import sys
import types

import numpy as np
import scipy.io as sio

REF = "/root/reference/scripts"
HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLES = (0, 1, 100, 1832)


def plain(o):
    """numpy scalars -> Python numbers (what roslibpy's JSON encoder sends over the wire)."""
    if isinstance(o, dict):
        return {k: plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [plain(v) for v in o]
    if isinstance(o, np.generic):
        return o.item()
    return o


def main():
    sys.modules["roslibpy"] = types.ModuleType("roslibpy")
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "matlab2ros"))
    import createbag as cb   # noqa: E402  (the reference's module)
    mat = sio.loadmat(os.path.join(REF, "data_IJAC2018.mat"))
    z, odo, u = (np.array(mat[k]) for k in ("observations", "odometry", "velocities"))
    head_l, head_o = cb.Header(), cb.Header()
    out = {}
    for t in range(z.shape[1]):               # the publishing loop of createbag.py:137-143, without the network
        laser = head_l.new_message(cb.mat2laser_scann(z[:, t]))
        odom = head_o.new_message(cb.mat2odometry(odo[:, t], u[:, t]))
        if t in SAMPLES:
            out[str(t)] = {"laser_scan": plain(laser), "odometry": plain(odom)}
    json.dump({"source": "scripts/matlab2ros/createbag.py:37-147 on scripts/data_IJAC2018.mat", "numpy_version": np.__version__,
               "samples": out}, open(os.path.join(HERE, "createbag_messages.json"), "w"), indent=1)
    print("wrote", len(out), "samples")


if __name__ == "__main__":
    main()
