from woft_amd.config import Config, load_config  # noqa: F401
