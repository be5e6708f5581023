"""Progress bar with the two reference extensions the epochers use
(ref whl:deepclustering2/tqdm/__init__.py:77-90)."""
from tqdm import tqdm as _tqdm

from deepclustering2.utils import flatten_dict, nice_dict


class tqdm(_tqdm):
    def __init__(self, iterable=None, desc=None, total=None, leave=False, ncols=2, dynamic_ncols=True, disable=None, **kwargs):
        import os
        if disable is None:
            disable = os.environ.get("MISEG_PROGRESS", "1") == "0"
        super().__init__(iterable, desc=desc, total=total, leave=leave, ncols=ncols, dynamic_ncols=dynamic_ncols, disable=disable,
                         bar_format="{l_bar}{bar}| {n_fmt}/{total_fmt} [{rate_fmt}{postfix}]", **kwargs)

    def set_desc_from_epocher(self, epocher):
        des = f"{epocher.__class__.__name__:<15} {epocher._cur_epoch:03d}"
        self.set_description(desc=des)
        return self

    def set_postfix_dict(self, dictionary):
        self._post_dict = flatten_dict(dictionary)
        self.set_postfix({k: f"{v:.3g}" if isinstance(v, float) else v for k, v in self._post_dict.items()})

    def _print_description(self):
        if getattr(self, "_post_dict", None) and not self.disable:
            print(f"{self.desc}: {nice_dict(self._post_dict)}")
