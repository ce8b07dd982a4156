from woft_amd.tracker import YAOFTrackerSingleControl, make_forward_compatible  # noqa: F401
