"""Minimal property tree with the aura-props calls the hot path makes (getNode + typed
getters/setters on PropertyNode).  Used only when the real `props` package is not
installed; with it installed the reference's own global tree is shared (drop-in)."""


class PropertyNode(object):
    def hasChild(self, name):
        return name in self.__dict__

    def getChild(self, path, create=False):
        node = self
        for tok in [t for t in str(path).split('/') if t]:
            nxt = node.__dict__.get(tok)
            if isinstance(nxt, PropertyNode):
                node = nxt
            elif create:
                nxt = PropertyNode()
                node.__dict__[tok] = nxt
                node = nxt
            else:
                return None
        return node

    def getChildren(self, expand=True):
        return sorted(self.__dict__.keys())

    # enumerated values ------------------------------------------------------
    def getLen(self, name):
        v = self.__dict__.get(name)
        return len(v) if isinstance(v, list) else 0

    def setLen(self, name, size, init_val=None):
        v = self.__dict__.get(name)
        if not isinstance(v, list):
            v = []
        fill = 0.0 if init_val is None else init_val
        v.extend([fill] * (size - len(v)))
        del v[size:]
        self.__dict__[name] = v

    def getFloatEnum(self, name, index):
        v = self.__dict__.get(name)
        return float(v[index]) if isinstance(v, list) and index < len(v) else 0.0

    def setFloatEnum(self, name, index, val):
        v = self.__dict__.get(name)
        if not isinstance(v, list):
            v = []
            self.__dict__[name] = v
        v.extend([0.0] * (index + 1 - len(v)))
        v[index] = float(val)

    # scalars ---------------------------------------------------------------
    def _scalar(self, name):
        v = self.__dict__.get(name)
        return None if isinstance(v, (PropertyNode, list)) else v

    def getFloat(self, name):
        v = self._scalar(name)
        try:
            return float(v) if v is not None else 0.0
        except ValueError:
            return 0.0

    def getInt(self, name):
        v = self._scalar(name)
        try:
            return int(float(v)) if v is not None else 0
        except ValueError:
            return 0

    def getString(self, name):
        v = self._scalar(name)
        return "" if v is None else str(v)

    def getBool(self, name):
        v = self._scalar(name)
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
