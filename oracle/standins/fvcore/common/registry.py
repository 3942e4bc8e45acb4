"""String -> class registry (decorator or direct register, get)."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, name
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map
