O=gpurun_out/kb5; mkdir -p $O
{
echo "=== OLD"; timeout 60 tools/bin/kbench_old 512 10000 512 | grep -v "fresh\|scale"
for t in 256 512; do echo "=== NEW $t"; timeout 60 tools/bin/kbench_a0 512 10000 $t | grep -v "fresh\|scale"; done
echo "=== NEW 256 between=1"; timeout 60 tools/bin/kbench_a0 512 10000 256 0 1 1 | grep -v "fresh\|scale"
echo "=== PC 256"; timeout 60 tools/bin/kbench_pc 512 10000 256 | grep -v "fresh\|scale"
for d in 1 2 3 4 5; do echo "=== dist $d"; timeout 60 tools/bin/kbench_a0 512 10000 0 $d | grep -v "fresh"; done
echo "=== shapes"; for sh in "64 10000" "512 8192" "512 1000" "300 37" "64 30000" "16 65536" "64 2049" "700 1" "100 2" "128 16384" "128 16385"; do timeout 60 tools/bin/kbench_a0 $sh | grep -v "fresh\|scale"; done
for d in 1 2 3 4 5; do echo "=== small dist $d"; timeout 60 tools/bin/kbench_a0 256 777 0 $d | grep -v "fresh\|scale"; timeout 60 tools/bin/kbench_a0 64 50000 0 $d | grep -v "fresh\|scale";  timeout 60 tools/bin/kbench_a0 64 20000 0 $d | grep -v "fresh\|scale"; done
} > $O/kb.log 2>&1
cat $O/kb.log
