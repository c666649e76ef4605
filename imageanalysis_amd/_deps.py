"""Late binding of the pieces of the reference environment the hot-path modules talk to.

Inside the reference tree (scripts/ on sys.path, aura-props installed) the real modules are
used so that state -- the global property tree, the message log, smart.json -- is shared with
the rest of process.py.  Outside it (tests, bench) small equivalents from hostlib/ are used."""
import importlib

try:                                    # third-party aura-props
    from props import getNode, PropertyNode   # noqa: F401
    HAVE_PROPS = True
except ImportError:
    from .hostlib.props_compat import getNode, PropertyNode   # noqa: F401
    HAVE_PROPS = False


_resolved = {}


def _ref_or_host(name):
    if name in _resolved:                # called per log line: resolve once
        return _resolved[name]
    mod = None
    if HAVE_PROPS:
        try:
            mod = importlib.import_module('lib.' + name)
        except Exception:
            mod = None
    if mod is None:
        try:
            mod = importlib.import_module('imageanalysis_amd.hostlib.' + name)
        except ImportError:
            mod = None
    _resolved[name] = mod
    return mod


def camera():
    return _ref_or_host('camera')


def logger():
    return _ref_or_host('logger')


def smart():
    """lib.smart inside the reference environment (per-pair surface / yaw estimates, SURVEY.md
    8f rank 2), else the deterministic half restated in imageanalysis_amd.smart."""
    if HAVE_PROPS:
        try:
            return importlib.import_module('lib.smart')
        except Exception:
            pass
    return importlib.import_module('imageanalysis_amd.smart')
