"""Plugin seam of the reference (src/util/import_helper.py:4-24): objects are built from a dotted path."""
import importlib


def import_from(module, obj_name):
    return getattr(importlib.import_module(module), obj_name)


def import_obj(s: str):
    module, _, obj_name = s.rpartition(".")
    return import_from(module, obj_name)
