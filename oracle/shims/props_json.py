"""ORACLE / TEST INFRASTRUCTURE ONLY.  Stand-in for ``props_json`` (aura-props):
just enough for /root/reference/scripts/lib/{project,smart}.py to import."""
import json

from props import PropertyNode


def _to_obj(node):
    out = {}
    for k, v in node.__dict__.items():
        out[k] = _to_obj(v) if isinstance(v, PropertyNode) else v
    return out


def _from_obj(node, obj):
    for k, v in obj.items():
        if isinstance(v, dict):
            child = PropertyNode()
            node.__dict__[k] = child
            _from_obj(child, v)
        else:
            node.__dict__[k] = v


def save(filename, node):
    with open(filename, 'w') as f:
        json.dump(_to_obj(node), f, indent=4, sort_keys=True)
    return True


def load(filename, node):
    try:
        with open(filename, 'r') as f:
            _from_obj(node, json.load(f))
        return True
    except (IOError, ValueError):
        return False
