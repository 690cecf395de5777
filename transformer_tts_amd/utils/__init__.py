"""``hparams`` config singleton -- the config surface of the reference path.

Mirrors the behaviour of the reference's ``utils/__init__.py:8-51``: a module-level object that
is configured ONCE from a ``.py`` file (every non-dunder global becomes an attribute), refuses to
be reconfigured, and raises ``AttributeError`` for unknown names.  Unlike the reference it does
not print every attribute it is asked for (``utils/__init__.py:16``).
"""
import importlib.util
import re
from pathlib import Path


class HParams:
    def __init__(self):
        object.__setattr__(self, "_configured", False)

    def is_configured(self):
        return self._configured

    def __getattr__(self, item):
        # only reached when normal lookup fails
        if not self.__dict__.get("_configured", False):
            raise AttributeError("HParams not configured yet. Call self.configure()")
        raise AttributeError(f'HParams does not have "{item}"')

    def configure(self, path):
        if self.is_configured():
            raise RuntimeError("Cannot reconfigure hparams!")
        path = Path(path).expanduser()
        if not path.exists():
            raise FileNotFoundError(f"Could not find hparams file {path}")
        if path.suffix != ".py":
            raise ValueError("`path` must be a python file")
        spec = importlib.util.spec_from_file_location("hparams", path)
        if spec is None:
            raise ValueError(f'could not load module from "{path}"')
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        dunder = re.compile(r"^__.+__$")
        for name, value in vars(mod).items():
            if dunder.match(name):
                continue
            if name in self.__dict__:
                raise AttributeError(f"module at `path` cannot contain attribute {name} as it overwrites "
                                     "an attribute of the same name in utils.hparams")
            setattr(self, name, value)
        self._configured = True


hparams = HParams()
