#!/bin/bash
# shipped kernel of round 3 (kb_ship) vs the tree's kernel without / with range hints (kb_head, KB_HINT=1), alternating
O=gpurun_out/${1:-hint}; mkdir -p $O
{
for rep in 1 2 3; do
  for rows in 512 64; do
    KB_EP=1 KB_UNIFORM=1 ./tools/kb/kb_ship $rows 10000 512 0 2>&1 | grep -E "between=" | sed "s/^/ship        rows=$rows rep=$rep: /"
    KB_EP=1 KB_UNIFORM=1 ./tools/kb/kb_head $rows 10000 512 0 2>&1 | grep -E "between=" | sed "s/^/head nohint rows=$rows rep=$rep: /"
    KB_HINT=1 KB_EP=1 KB_UNIFORM=1 ./tools/kb/kb_head $rows 10000 512 0 2>&1 | grep -E "between=" | sed "s/^/head hint   rows=$rows rep=$rep: /"
  done
done
echo "== every distribution, fresh draws (hit / miss counts) and steady state, hints on"
for dist in 0 1 2 3 4 5 6 7; do
  KB_HINT=1 KB_EP=1 ./tools/kb/kb_head 512 10000 512 $dist 2>&1 | grep -E "fresh draw|steady|between=" | sed "s/^/dist=$dist: /"
done
echo "== other row lengths, hints on (mismatches must be 0)"
for n in 1 7 100 2047 2049 4096 8192 20000 65536; do
  KB_HINT=1 ./tools/kb/kb_head 64 $n 0 0 2>&1 | grep -E "steady|between=" | sed "s/^/n=$n: /"
done
} 2>&1 | tee $O/range_hint.txt | cut -c1-230
