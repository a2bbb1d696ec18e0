from openpvsg_amd.unitrack import KalmanFilter, chi2inv95  # noqa: F401
