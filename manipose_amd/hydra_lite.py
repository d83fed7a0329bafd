"""The slice of Hydra's override grammar the reference's entry points are driven with (hydra-core is not a dependency here):
  group=name | +group=name   merge <conf>/<group>/<name>.yaml into that group (new keys allowed)
  group.key=value            set an EXISTING key (a typo is an error, as with Hydra)
  +group.key=value           add a new key
Scalars are read the way Hydra reads them (`4e-5` and `1.` are floats, True/False booleans).  Errors are SystemExit with a message."""
from __future__ import annotations

import os

import yaml


class Cfg(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    @staticmethod
    def wrap(d):
        return Cfg({k: Cfg.wrap(v) if isinstance(v, dict) else v for k, v in d.items()})


def parse_value(text):
    t = text.strip()
    for conv in (int, float):
        try:
            return conv(t)
        except ValueError:
            pass
    low = t.lower()
    if low in ("true", "false"):
        return low == "true"
    if low in ("null", "none", "~"):
        return None
    try:
        return yaml.safe_load(t)
    except yaml.YAMLError:
        return t


def _coerce(node):
    """YAML 1.1 leaves `1e-3` a string; Hydra / OmegaConf read it as a float."""
    if isinstance(node, dict):
        return {k: _coerce(v) for k, v in node.items()}
    if isinstance(node, str):
        try:
            return float(node) if any(c in node for c in "eE.") and node.strip() else node
        except ValueError:
            return node
    return node


def load_config(conf_dir, argv, extra_defaults=None):
    with open(os.path.join(conf_dir, "config.yaml")) as f:
        cfg = _coerce(yaml.safe_load(f))
    cfg.pop("hydra", None)
    for k, v in (extra_defaults or {}).items():
        grp, key = k.split(".")
        cfg[grp][key] = v
    for arg in argv:
        if "=" not in arg:
            raise SystemExit(f"expected `group.key=value` or `group=name` overrides, got {arg!r}")
        path, val = arg.split("=", 1)
        add = path.startswith("+")
        path = path.lstrip("+")
        if "." not in path:                                   # config group selection
            if path not in cfg:
                raise SystemExit(f"unknown config group {path!r} (groups: {sorted(cfg)})")
            fn = os.path.join(conf_dir, path, val + ".yaml")
            if not os.path.exists(fn):
                have = sorted(x[:-5] for x in os.listdir(os.path.join(conf_dir, path))) if os.path.isdir(os.path.join(conf_dir, path)) else []
                raise SystemExit(f"no config {path}/{val}.yaml (available for {path!r}: {have})")
            with open(fn) as f:
                cfg[path].update(_coerce(yaml.safe_load(f) or {}))
            continue
        grp, key = path.split(".", 1)
        if grp not in cfg:
            raise SystemExit(f"unknown config group {grp!r} (groups: {sorted(cfg)})")
        if key not in cfg[grp] and not add:
            raise SystemExit(f"unknown key {grp}.{key} (keys of {grp!r}: {sorted(cfg[grp])}); prefix with + to add a new key")
        cfg[grp][key] = parse_value(val)
    return Cfg.wrap(cfg)
