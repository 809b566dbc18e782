O=gpurun_out/kb7; mkdir -p $O
{
echo "=== OLD"; timeout 60 tools/bin/kbench_old 512 10000 512 | grep -v "fresh\|scale"
for t in 256 512; do echo "=== NEW $t"; timeout 60 tools/bin/kbench_a0 512 10000 $t | grep -v "fresh\|scale"; done
echo "=== NEW 256 between=1"; timeout 60 tools/bin/kbench_a0 512 10000 256 0 1 1 | grep -v "fresh\|scale"
echo "=== PC 256"; timeout 60 tools/bin/kbench_pc 512 10000 256 | grep -v "fresh\|scale"
for d in 1 2 3 4 5; do echo "=== dist $d"; timeout 60 tools/bin/kbench_a0 512 10000 256 $d | grep -v "fresh\|scale"; done
} > $O/kb.log 2>&1
cat $O/kb.log
