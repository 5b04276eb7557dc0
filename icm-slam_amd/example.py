"""Counterpart of the reference's `scripts/example.py`: the same driver loop
(scripts/example.py:37-54), offline.  The reference needs a live rosbridge; here the recorded
sequence named by `config.file` is loaded and `inicializar_offline()` runs the same
initialisation pass on it.

    python example.py [config.yaml] [data.mat|data.npz]
"""
import sys
from copy import deepcopy as copy

import numpy as np

from ICM_ROS import ICM_ROS
from ICM_SLAM_tools import ConfigICM, calc_cambio


class My_method(ICM_ROS):
    """Placeholders of the reference's example subclass (scripts/example.py:13-35): methods with a
    trailing underscore are never called.  Overriding g/h/fun_x/fun_xn themselves is refused by
    the HIP build (Python callbacks cannot run inside the kernels)."""

    def __init__(self, config):
        ICM_ROS.__init__(self, config)


if __name__ == '__main__':
    config = ConfigICM(sys.argv[1] if len(sys.argv) > 1 else 'config_default.yaml')
    ICM = ICM_ROS(config)
    ICM.load_data(sys.argv[2] if len(sys.argv) > 2 else None)
    ICM.inicializar_offline()
    if ICM.iterations_flag:
        mapa_viejo = copy(ICM.mapa_viejo)
        x = copy(ICM.positions)
        for iteracionICM in range(config.N):
            print('iteración ICM : ', iteracionICM + 1)
            mapa_refinado, x = ICM.iterations_process_offline(mapa_viejo, x)
            print('Correccion: ', np.linalg.norm(x - ICM.positions, axis=1).sum(),
                  ' cambio (min,max,medio): ', calc_cambio(mapa_refinado, mapa_viejo))
            mapa_viejo = copy(mapa_refinado)  # as scripts/ICM_ROS.py:311
