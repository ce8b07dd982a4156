from woft_amd.flow_provider import RAFTWrapper  # noqa: F401
