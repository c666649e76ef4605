"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Minimal stand-in for the third-party ``props`` package (aura-props, named in
the reference's environment.yml:44) so that /root/reference/scripts/lib/*.py can
be imported in this container to generate golden vectors (tools: oracle/gen_golden.py).
Only the calls the reference's matcher/optimizer/image/camera modules make are
provided: getNode(path, create), and the PropertyNode getters/setters.
"""


class PropertyNode(object):
    def __init__(self):
        pass

    # -- tree navigation ------------------------------------------------
    def hasChild(self, name):
        return name in self.__dict__

    def getChild(self, path, create=False):
        node = self
        for tok in [t for t in path.split('/') if t != '']:
            if tok in node.__dict__ and isinstance(node.__dict__[tok], PropertyNode):
                node = node.__dict__[tok]
            elif create:
                child = PropertyNode()
                node.__dict__[tok] = child
                node = child
            else:
                return None
        return node

    def getChildren(self, expand=True):
        return sorted(self.__dict__.keys())

    def isLeaf(self, name):
        return name in self.__dict__ and not isinstance(self.__dict__[name], PropertyNode)

    # -- enumerated (list) values ----------------------------------------
    def getLen(self, name):
        v = self.__dict__.get(name)
        return len(v) if isinstance(v, list) else 0

    def setLen(self, name, size, init_val=None):
        v = self.__dict__.get(name)
        if not isinstance(v, list):
            v = []
        while len(v) < size:
            v.append(init_val if init_val is not None else 0.0)
        del v[size:]
        self.__dict__[name] = v

    def getFloatEnum(self, name, index):
        v = self.__dict__.get(name)
        if isinstance(v, list) and index < len(v):
            return float(v[index])
        return 0.0

    def setFloatEnum(self, name, index, val):
        v = self.__dict__.get(name)
        if not isinstance(v, list):
            v = []
            self.__dict__[name] = v
        while len(v) <= index:
            v.append(0.0)
        v[index] = float(val)

    # -- scalars -------------------------------------------------------
    def getFloat(self, name):
        v = self.__dict__.get(name)
        if v is None or isinstance(v, (PropertyNode, list)):
            return 0.0
        try:
            return float(v)
        except ValueError:
            return 0.0

    def getInt(self, name):
        v = self.__dict__.get(name)
        if v is None or isinstance(v, (PropertyNode, list)):
            return 0
        try:
            return int(float(v))
        except ValueError:
            return 0

    def getString(self, name):
        v = self.__dict__.get(name)
        if v is None or isinstance(v, (PropertyNode, list)):
            return ""
        return str(v)

    def getBool(self, name):
        v = self.__dict__.get(name)
        if isinstance(v, str):
            return v.lower() in ('true', '1')
        return bool(v)

    def setFloat(self, name, val):
        self.__dict__[name] = float(val)

    def setInt(self, name, val):
        self.__dict__[name] = int(val)

    def setString(self, name, val):
        self.__dict__[name] = str(val)

    def setBool(self, name, val):
        self.__dict__[name] = bool(val)


root = PropertyNode()


def getNode(path, create=False):
    if path in ('', '/'):
        return root
    return root.getChild(path, create)
