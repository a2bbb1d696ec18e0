"""Clean-room equivalent of the part of `mmcv.Config` the reference uses (tools/test.py:141-150,
models/mask2former/mask2former_head.py:83-92): python-file configs with `_base_` inheritance and
recursive dict merge, attribute access, `.get`, `.update`, deepcopy, `merge_from_dict`, and the
`DictAction` argparse action for `--cfg-options k=v`."""
import argparse
import ast
import copy
import os
import types

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def copy(self):
        return ConfigDict(self)


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


def _merge_a_into_b(a, b):
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and isinstance(b[k], dict) and not v.get(DELETE_KEY, False):
            b[k] = _merge_a_into_b(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = copy.deepcopy(v)
    return b


def _load_py(path):
    with open(path, 'r') as f:
        src = f.read()
    ns = {'__file__': path}
    exec(compile(src, path, 'exec'), ns)
    return {k: v for k, v in ns.items()
            if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType))}


def _file2dict(path):
    path = os.path.abspath(os.path.expanduser(path))
    if not os.path.isfile(path):
        raise FileNotFoundError('config file not found: %s' % path)
    if not path.endswith('.py'):
        raise IOError('Only py type configs are supported')
    cfg = _load_py(path)
    if BASE_KEY in cfg:
        bases = cfg.pop(BASE_KEY)
        bases = bases if isinstance(bases, list) else [bases]
        merged = {}
        for b in bases:
            bd = _file2dict(os.path.join(os.path.dirname(path), b))
            dup = set(merged) & set(bd)
            if dup:
                raise KeyError('Duplicate key is not allowed among bases: %s' % sorted(dup))
            merged.update(bd)
        cfg = _merge_a_into_b(cfg, merged)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename, **kwargs):
        return Config(_file2dict(filename), filename=str(filename))

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def items(self):
        return self._cfg_dict.items()

    def pop(self, key, default=None):
        return self._cfg_dict.pop(key, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options):
        """`a.b.c=v` style overrides (tools/test.py:149-150)."""
        nested = {}
        for full, v in options.items():
            d = nested
            parts = full.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge_a_into_b(nested, self._cfg_dict)))

    def __deepcopy__(self, memo):
        return Config(copy.deepcopy(dict(self._cfg_dict), memo), self._filename)


class DictAction(argparse.Action):
    """`--cfg-options a.b=1 c=[1,2] d=x` -> {'a.b': 1, 'c': [1, 2], 'd': 'x'} (tools/test.py:86-93)."""

    @staticmethod
    def _parse(val):
        if ',' in val and not val.startswith(('[', '(', '{', '"', "'")):
            return [DictAction._parse(v) for v in val.split(',')]
        try:
            return ast.literal_eval(val)
        except (ValueError, SyntaxError):
            if val.lower() in ('true', 'false'):
                return val.lower() == 'true'
            if val.lower() in ('none', 'null'):
                return None
            if ',' in val and not val.startswith(('[', '(')):
                return [DictAction._parse(v) for v in val.split(',')]
            return val

    def __call__(self, parser, namespace, values, option_string=None):
        options = {}
        for kv in values:
            k, v = kv.split('=', maxsplit=1)
            options[k] = self._parse(v)
        setattr(namespace, self.dest, options)
