"""ORACLE / TEST INFRASTRUCTURE ONLY.  Import-time stand-in for NavPy; the hot
path (matcher/optimizer) never calls into it."""


def lla2ned(*a, **k):
    raise NotImplementedError("navpy stand-in")


def ned2lla(*a, **k):
    raise NotImplementedError("navpy stand-in")
