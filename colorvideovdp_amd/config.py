"""Configuration lookup, mirroring pycvvdp/utils.py:133-174 (`config_files.find`, `json2dict`).

Search order for a configuration file `<stem>.json`:
  1. a file in `config_paths` whose basename starts with `<stem>` and ends with `.json`
  2. `<dir>/<stem>.json` for every directory in `config_paths`
  3. `$CVVDP_PATH/<stem>.json`
  4. the built-in bundle `colorvideovdp_amd/data/vvdp_data.json` (section `<stem>`), which holds the
     calibration values of ColorVideoVDP v0.5.6 in this project's own single-file layout.
Files found in 1-3 use the reference's file formats, so existing custom displays keep working.
"""
import json
import os

_BUNDLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "vvdp_data.json")
_bundle = None


def _builtin():
    global _bundle
    if _bundle is None:
        with open(_BUNDLE_PATH) as f:
            _bundle = json.load(f)
    return _bundle


class config_files:
    @classmethod
    def find(cls, fname, config_paths):
        """Returns a path, or the string 'builtin:<stem>' when only the bundle has the section."""
        if not isinstance(config_paths, list):
            raise RuntimeError("config_paths must be a list")
        stem, ext = os.path.splitext(fname)
        for cp in config_paths:
            if not (os.path.isfile(cp) or os.path.isdir(cp)):
                raise RuntimeError(f"config_path '{cp}' does not exist")
            base = os.path.basename(cp)
            if os.path.isfile(cp) and base.startswith(stem) and base.endswith(ext):
                return cp
        for cp in config_paths:
            if os.path.isdir(cp) and os.path.isfile(os.path.join(cp, fname)):
                return os.path.join(cp, fname)
        env = os.getenv("CVVDP_PATH")
        if env is not None and os.path.isfile(os.path.join(env, fname)):
            return os.path.join(env, fname)
        if stem in _builtin():
            return "builtin:" + stem
        raise RuntimeError(f"The configuration file {fname} not found")


def json2dict(path):
    if path.startswith("builtin:"):
        return _builtin()[path[len("builtin:"):]]
    if not os.path.isfile(path):
        raise RuntimeError(f"Error: Cannot find file {path}.")
    with open(path) as f:
        return json.load(f)


def load_config(fname, config_paths):
    return json2dict(config_files.find(fname, config_paths))
