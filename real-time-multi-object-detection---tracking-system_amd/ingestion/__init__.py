from .reader import FrameReader, RTSPReader, RawVideoCapture, SyntheticCapture, register_backend  # noqa: F401
