from woft_amd.homography import compose_H  # noqa: F401
