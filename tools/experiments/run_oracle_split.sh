#!/bin/bash
# kb_ship vs kb_oracle, alternating, bench shape (512 rows) and N=8 per-GPU shape (64 rows); report-path epilogue on
O=gpurun_out/${1:-oracle}; mkdir -p $O
for rep in 1 2 3; do
  for rows in 512 64; do
    for b in kb_ship kb_oracle; do
      KB_EP=1 KB_UNIFORM=1 ./tools/kb/$b $rows 10000 512 0 2>&1 | grep -E "steady|between=" | sed "s/^/$b rows=$rows rep=$rep: /"
    done
  done
done | tee $O/oracle_split.txt
