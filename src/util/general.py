"""Small helpers of the reference's src/util/general.py plus the attribute-dict the renderer returns.

The reference returns `dotmap.DotMap` objects from NeRFRendererDGS.forward (nerf_renderer.py:421-430); callers only
use attribute access (`out.fine.rgb`, diner.py:87-88), so a minimal attribute dict is a faithful stand-in and avoids
the third-party dependency (dotmap is used when it is installed)."""
import json

try:                                    # pragma: no cover - depends on the environment
    from dotmap import DotMap           # noqa: F401
except Exception:                       # dotmap is not installed in the MI355X image
    class DotMap(dict):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

        def toDict(self):
            return {k: (v.toDict() if isinstance(v, DotMap) else v) for k, v in self.items()}


def prefix_dict_keys(d, prefix):
    return {prefix + k: v for k, v in d.items()}


def load_json(path):
    with open(path) as f:
        return json.load(f)
