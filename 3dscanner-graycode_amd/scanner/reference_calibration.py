"""Calibration constants that ship with the reference as data files
(data/calib_results/{cam_1080,cam_1440,proj}/*.npy); used as benchmark / example constants
(last row: they are data, not code).  float64 literals round-trip exactly."""
import numpy as np

CAM_MTX = np.array([[1276.2366469651317, 0.0, 967.1092076214613], [0.0, 1276.2366469651317, 522.6606151959928], [0.0, 0.0, 1.0]], dtype=np.float64)

CAM_DIST = np.array([[-0.03734052955485712], [-0.10770214128917989], [-7.562558465029136e-05], [0.0012286579663293728], [0.027111816245413922]], dtype=np.float64)

CAM1440_MTX = np.array([[1799.4694793738233, 0.0, 1246.519484046554], [0.0, 1799.4694793738233, 680.3290643970407], [0.0, 0.0, 1.0]], dtype=np.float64)

CAM1440_DIST = np.array([[-0.5138336530315805], [0.38134494155048765], [0.005218865792851247], [0.005120239190266125], [-0.16170378586244424]], dtype=np.float64)

PROJ_MTX = np.array([[2648.8646560955926, 0.0, 543.349681479279], [0.0, 2400.342677727396, 272.1757379152153], [0.0, 0.0, 1.0]], dtype=np.float64)

PROJ_DIST = np.array([[-0.27538871785643054, 6.681621239710083, -0.0014415703322899751, -0.041959078402546944, -31.599309381926286]], dtype=np.float64)
