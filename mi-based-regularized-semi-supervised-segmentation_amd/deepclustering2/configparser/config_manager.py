"""YAML defaults + ``a.b.c=value`` CLI overrides (ref whl:deepclustering2/configparser/config_manager.py:10-54,
_yaml_parser.py:17-121).  ``python main.py Trainer.name=udaiic Optim.lr=1e-7 --config_path other.yaml``."""
import argparse
from copy import deepcopy
from pathlib import Path
from typing import Any, Dict, List, Optional

import yaml

from deepclustering2.utils import dict_merge


def parse_override(string: str) -> Dict[str, Any]:
    """'a.b=1' / 'a.b:1' / 'a.b:!int=1' -> {'a': {'b': 1}}"""
    if string == "":
        return {}
    if "!" in string:
        string = string.replace(":", ": ").replace("=", " ").replace("!", " !!")
    else:
        string = string.replace(":", ": ").replace("=", ": ")
    flat = yaml.safe_load(string)
    if not isinstance(flat, dict) or len(flat) != 1:
        return {}
    (key, value), = flat.items()
    for k in reversed(str(key).split(".")):
        value = {k: value}
    return value


class ConfigManger:
    def __init__(self, DEFAULT_CONFIG_PATH: str = None, verbose=True, argv: Optional[List[str]] = None) -> None:
        parser = argparse.ArgumentParser("yaml config with key=value overrides")
        parser.add_argument("--config_path", type=str, default=None)
        parser.add_argument("strings", nargs="*", type=str, default=[""])
        args, _ = parser.parse_known_args(argv)
        self._parsed: Dict[str, Any] = {}
        for s in args.strings:
            dict_merge(self._parsed, parse_override(s))
        path = args.config_path or DEFAULT_CONFIG_PATH
        self._default = yaml.safe_load(open(str(Path(path)))) if path else {}
        self._merged = dict_merge(deepcopy(self._default), deepcopy(self._parsed))
        if verbose:
            print(yaml.dump(self._merged))

    @property
    def default_config(self):
        return deepcopy(self._default)

    @property
    def parsed_config(self):
        return deepcopy(self._parsed)

    @property
    def merged_config(self):
        return deepcopy(self._merged)

    @property
    def config(self):
        return self.merged_config
