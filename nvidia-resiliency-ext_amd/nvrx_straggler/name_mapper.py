"""Name <-> dense integer id tables that are identical on every rank.

Behavioural contract taken from the reference's ``NameMapper`` (name_mapper.py:22-161) and pinned by
its tests (tests/straggler/unit/test_name_mapper.py:64-98,231-236):

* ids are consecutive from 0, separately for sections and kernels, and never change once given;
* when any rank meets an unknown name, all ranks exchange their current name lists with ONE
  ``all_gather_object`` and assign ids by walking the gathered lists rank-major -- every rank's
  sections first, then every rank's kernels (name_mapper.py:71-81);
* otherwise no string ever crosses the wire.

The "does any rank have a new name" check is NOT a separate collective here: the flag rides in the
last word of each rank's exchange row (``ReportGenerator`` reads it back from the score kernel's
metadata), so ``gather_and_assign_ids`` keeps its reference signature but the hot path calls
``has_all_names`` + ``sync_names`` instead.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

from .dist_utils import all_gather_object, is_all_true


class NameMapper:
    def __init__(self, pg=None):
        self.group = pg
        self.kernel_name_to_id: Dict[str, int] = {}
        self.id_to_kernel_name: Dict[int, str] = {}
        self.section_name_to_id: Dict[str, int] = {}
        self.id_to_section_name: Dict[int, str] = {}
        self.kernel_counter: int = 0
        self.section_counter: int = 0
        #: bumped whenever an id is added; lets callers cache anything derived from the tables
        self.version: int = 0

    # ---- queries -------------------------------------------------------------------------------
    def has_all_names(self, kernel_names: Iterable[str], section_names: Iterable[str]) -> bool:
        k2i, s2i = self.kernel_name_to_id, self.section_name_to_id
        return all(n in k2i for n in kernel_names) and all(n in s2i for n in section_names)

    # reference spelling (name_mapper.py:46-52)
    _check_if_has_all_names = has_all_names

    def get_kernel_name(self, kernel_id: int) -> str:
        return self.id_to_kernel_name[kernel_id]

    def get_kernel_id(self, kernel_name: str) -> int:
        return self.kernel_name_to_id[kernel_name]

    def get_section_name(self, section_id: int) -> str:
        return self.id_to_section_name[section_id]

    def get_section_id(self, section_name: str) -> int:
        return self.section_name_to_id[section_name]

    # ---- id assignment -------------------------------------------------------------------------
    def _assign_kernel_id(self, kernel_name: str) -> int:
        idx = self.kernel_name_to_id.get(kernel_name)
        if idx is None:
            idx = self.kernel_counter
            self.kernel_name_to_id[kernel_name] = idx
            self.id_to_kernel_name[idx] = kernel_name
            self.kernel_counter += 1
            self.version += 1
        return idx

    def _assign_section_id(self, section_name: str) -> int:
        idx = self.section_name_to_id.get(section_name)
        if idx is None:
            idx = self.section_counter
            self.section_name_to_id[section_name] = idx
            self.id_to_section_name[idx] = section_name
            self.section_counter += 1
            self.version += 1
        return idx

    def sync_names(self, kernel_names: List[str], section_names: List[str]) -> None:
        """Cold path: exchange name lists (one all_gather_object) and extend the tables."""
        gathered = all_gather_object((list(section_names), list(kernel_names)), self.group)
        for sections, _ in gathered:
            for name in sections:
                self._assign_section_id(name)
        for _, kernels in gathered:
            for name in kernels:
                self._assign_kernel_id(name)

    def gather_and_assign_ids(self, kernel_names: List[str], section_names: List[str]) -> None:
        """Reference-compatible entry point (name_mapper.py:54-81): flag all-reduce, then sync if
        any rank saw a new name.  Collective: must be called by every rank of the group."""
        everyone_ok = is_all_true(self.has_all_names(kernel_names, section_names), self.group)
        if not everyone_ok:
            self.sync_names(kernel_names, section_names)
