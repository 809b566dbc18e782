"""Name <-> dense integer id tables that are identical on every rank.

Behavioural contract taken from the reference's ``NameMapper`` (name_mapper.py:22-161) and pinned by
its tests (tests/straggler/unit/test_name_mapper.py:64-98,231-236):

* ids are consecutive from 0, separately for sections and kernels, and never change once given;
* when any rank meets an unknown name, all ranks exchange their current name lists with ONE
  ``all_gather_object`` and assign ids by walking the gathered lists rank-major -- every rank's
  sections first, then every rank's kernels (name_mapper.py:71-81);
* otherwise no string ever crosses the wire.

Wire format of that cold exchange.  With per-kernel tracing a rank meets thousands of
kernel keys on its first report and a hipBLASLt key is ~600 bytes, so pickling every name to every rank is
megabytes.  Ranks therefore send only the names that have no id yet (the walk below skips known names anyway, so
the ids come out the same), and a rank with ``BULK_NAMES`` or more new kernel names sends their 8-byte BLAKE2b
digests instead of the strings.  SPMD ranks launch the same kernels: every rank then already holds the string of
every digest and nothing else is sent.  Only digests that some rank cannot resolve cost a second
``all_gather_object`` in which the lowest rank that owns each such name supplies it.  Sections (few, short, and
rank 0 needs the names of sections it never ran) always travel as strings.  ``NVRX_DEBUG_NAME_EXCHANGE=strings`` forces
strings everywhere, ``=digests`` digests for any number of names.  Two different names with one digest on a rank
raise; two ranks holding different names with one digest and no rank holding both cannot be told apart
(probability ~1e-11 at 10^4 names) -- use ``strings`` if that matters.

The "does any rank have a new name" check is NOT a separate collective here: the flag rides in the
last word of each rank's exchange row (``ReportGenerator`` reads it back from the score kernel's
metadata), so ``gather_and_assign_ids`` keeps its reference signature but the hot path calls
``has_all_names`` + ``sync_names`` instead.
"""
from __future__ import annotations

import hashlib
import os
from typing import Dict, Iterable, List

import numpy as np

from .dist_utils import all_gather_object, get_rank, is_all_true

BULK_NAMES = 32  # new kernel names on a rank from which it sends digests instead of strings


def name_digest(name: str) -> int:
    """64-bit BLAKE2b digest of a name: the same on every rank and in every process (unlike ``hash``)."""
    return int.from_bytes(hashlib.blake2b(name.encode("utf-8", "surrogatepass"), digest_size=8).digest(), "little")


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
        #: digest -> kernel name, for every kernel name this rank has seen (collision check + resolution)
        self._digest_to_kernel: Dict[int, str] = {}

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

    def _remember_digest(self, name: str) -> int:
        d = name_digest(name)
        known = self._digest_to_kernel.setdefault(d, name)
        if known != name:
            raise RuntimeError(
                f"64-bit name digest collision between kernel names {known!r} and {name!r}; "
                "set NVRX_DEBUG_NAME_EXCHANGE=strings")
        return d

    def sync_names(self, kernel_names: List[str], section_names: List[str]) -> None:
        """Cold path: exchange the names that have no id yet (one ``all_gather_object``; a second one only when
        some rank cannot resolve a digest) and extend the tables.  Ids are assigned in order of first appearance,
        rank-major, sections before kernels -- exactly the walk of the reference (name_mapper.py:71-81)."""
        new_sections = [n for n in section_names if n not in self.section_name_to_id]
        new_kernels = [n for n in kernel_names if n not in self.kernel_name_to_id]
        mode = os.environ.get("NVRX_DEBUG_NAME_EXCHANGE", "auto")
        if mode == "strings":
            # the reference's exchange, verbatim (no digest is ever computed: the escape hatch for a digest collision);
            # the variable must be set on every rank
            gathered = all_gather_object((new_sections, new_kernels), self.group)
            for sections, _ in gathered:
                for name in sections:
                    self._assign_section_id(name)
            for _, kernels in gathered:
                for name in kernels:
                    self._assign_kernel_id(name)
            return
        as_digests = mode == "digests" or len(new_kernels) >= BULK_NAMES
        my_digests = [self._remember_digest(n) for n in new_kernels]
        if as_digests:
            payload = ("d", np.asarray(my_digests, dtype=np.uint64).tobytes())
        else:
            payload = ("s", new_kernels)
        gathered = all_gather_object((new_sections, payload), self.group)

        for sections, _ in gathered:
            for name in sections:
                self._assign_section_id(name)

        world = len(gathered)
        order: List[int] = []           # new digests in order of first appearance, rank-major
        owners: Dict[int, int] = {}     # digest -> lowest rank that holds the name
        n_owners: Dict[int, int] = {}   # digest -> how many ranks hold it
        spelled = set()                 # digests whose string some rank sent in this round
        for r, (_, (kind, data)) in enumerate(gathered):
            if kind == "s":
                digests = [self._remember_digest(n) for n in data]
                spelled.update(digests)
            else:
                digests = np.frombuffer(data, dtype=np.uint64).tolist()
            for d in dict.fromkeys(digests):
                if d not in owners:
                    owners[d] = r
                    n_owners[d] = 0
                    order.append(d)
                n_owners[d] += 1
        # every rank computes the same list: digests nobody spelled out and not every rank holds
        unresolved = [d for d in order if d not in spelled and n_owners[d] < world]
        if unresolved:
            me = get_rank(self.group)
            mine = {d: self._digest_to_kernel[d] for d in unresolved if owners[d] == me}
            for part in all_gather_object(mine, self.group):
                for name in part.values():
                    self._remember_digest(name)
        for d in order:
            self._assign_kernel_id(self._digest_to_kernel[d])

    def gather_and_assign_ids(self, kernel_names: List[str], section_names: List[str]) -> None:
        """Reference-compatible entry point (name_mapper.py:54-81): flag all-reduce, then sync if
        any rank saw a new name.  Collective: must be called by every rank of the group."""
        everyone_ok = is_all_true(self.has_all_names(kernel_names, section_names), self.group)
        if not everyone_ok:
            self.sync_names(kernel_names, section_names)
