"""Attribute-dict stand-in for yacs.config.CfgNode.

Used ONLY by tools/gen_golden.py in the build container to import the read-only
reference (`/root/reference/voxelnet/config.py:1` does `from yacs.config import
CfgNode`; yacs is not installed here).  The reference hot path uses the node as
a nested attribute dict with .clone() — nothing else.
Contains no reference code.  Never shipped to / used on the GPU box.
"""
import copy


class CfgNode(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        pass
