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
    assert name not in REPLACED, name
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


# Modules of scripts/lib this package REPLACES (INTEGRATION.md section 2 puts a shim file of the
# same name in their place).  Nothing here may bind the reference's implementation of one of
# them: round 4 bound lib.smart inside the reference environment, which sent every matching pair
# of find_matches through cv2.triangulatePoints + estimateAffinePartial2D on the host instead of
# the batched device kernels.  tests/test_dropin.py keeps this true.
REPLACED = ('matcher', 'optimizer', 'smart', 'match_cleanup', 'groups')


def smart():
    """imageanalysis_amd.smart, in every environment: the per-pair surface / yaw estimates of
    scripts/lib/smart.py with the triangulation and the similarity fit on the device.  Inside the
    reference environment it works on the reference's own /smart property tree (getNode above is
    props.getNode) and reads / writes smart.json through props_json, so process.py:219-221,
    239-240 and lib/match_cleanup.py:309-316 see the same state as with lib/smart.py."""
    return importlib.import_module('imageanalysis_amd.smart')


def srtm():
    """lib.srtm (SRTM tile download + interpolation, scripts/lib/srtm.py) when the reference
    environment is there, else None.  Out of scope here (network + tile cache); the smart mirror
    delegates update_srtm_elevations() to it exactly like the reference does."""
    if 'srtm' not in _resolved:
        mod = None
        if HAVE_PROPS:
            try:
                mod = importlib.import_module('lib.srtm')
            except Exception:
                mod = None
        _resolved['srtm'] = mod
    return _resolved['srtm']
