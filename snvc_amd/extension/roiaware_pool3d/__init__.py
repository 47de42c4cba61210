from . import roiaware_pool3d_utils  # noqa: F401
