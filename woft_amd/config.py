"""Config system with the reference's semantics (/root/reference/pytracking/utils/config.py:5-43):
a `Config` is an attribute bag whose missing attributes read as a falsy, empty `Config`, so every
`if C.foo.bar:` is a feature flag that defaults to off; `load_config(path)` executes a python
module and returns its `get_config()`."""
import importlib.util
import logging

logger = logging.getLogger(__name__)


class Config:
    def __getattr__(self, name):
        # only reached when normal lookup fails: C.a.b.c never raises
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return Config()

    def __bool__(self):
        return False

    def merge(self, other, update_dicts=False):
        for key, value in other.__dict__.items():
            if key in self.__dict__:
                cur = self.__dict__[key]
                if update_dicts and isinstance(cur, dict) and isinstance(value, dict):
                    cur.update(value)
                    continue
                logger.debug(f"Rewriting key [{key}] in config. ({cur} -> {value})")
            setattr(self, key, value)

    def __repr__(self):
        return repr(self.__dict__)


def load_config(path):
    spec = importlib.util.spec_from_file_location("tracker_config", str(path))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module.get_config()
