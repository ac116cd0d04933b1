"""name -> class registry with the semantics of BasicSR's `Registry`
(HAT/ESC/basicsr/utils/registry.py:4-88): `register()` as decorator or call, duplicate names are an
error, `get(name)` falls back to `name + '_' + suffix` and raises KeyError when absent.

When the real `basicsr` package is importable its `ARCH_REGISTRY` / `MODEL_REGISTRY` objects are
used instead, so that `build_network({'type': 'HAT', ...})` (basicsr/archs/__init__.py:18-24) finds
this implementation; otherwise build-local registries with the same behaviour are used.
"""
from __future__ import annotations


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj, suffix=None):
        if isinstance(suffix, str):
            name = f"{name}_{suffix}"
        if name in self._obj_map:
            raise AssertionError(f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = obj

    def register(self, obj=None, suffix=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class, suffix)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj, suffix)

    def get(self, name, suffix='basicsr'):
        ret = self._obj_map.get(name)
        if ret is None:
            ret = self._obj_map.get(f"{name}_{suffix}")
            if ret is not None:
                print(f'Name {name} is not found, use name: {name}_{suffix}!')
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


def _resolve():
    try:  # pragma: no cover - basicsr is not installed in the build image
        from basicsr.utils.registry import ARCH_REGISTRY as A, MODEL_REGISTRY as M
        return A, M, True
    except Exception:
        return Registry('arch'), Registry('model'), False


ARCH_REGISTRY, MODEL_REGISTRY, USING_BASICSR = _resolve()


def build_network(opt: dict):
    """`basicsr.archs.build_network` (archs/__init__.py:18-24): pop 'type', instantiate from the registry."""
    opt = dict(opt)
    network_type = opt.pop('type')
    from . import archs  # noqa: F401  (registers HAT / HATX, as basicsr's arch scan does on import)
    return ARCH_REGISTRY.get(network_type)(**opt)
