cd /root/repo
export NVRX_KTRACE_DEBUG=1
echo "--- warm the SDK once (cold start-up cost)"; ( time timeout -s KILL 150 python tools/debug/ktrace_probe.py > /tmp/warm.log 2>&1 ) 2>&1 | grep real; grep -c "probe" /tmp/warm.log
echo "--- A: plain process holding an idle HIP context for 80 s"
python -c "
import torch, time
x = torch.randn(1024, 1024, device='cuda'); torch.cuda.synchronize(); print('A up', flush=True); time.sleep(80)" &
APID=$!
sleep 15
echo "--- B: tracer process while A is alive"; ( time timeout -s KILL 60 python tools/debug/ktrace_probe.py > /tmp/b.log 2>&1 ) 2>&1 | grep real; grep "nvrx_ktrace\|probe" /tmp/b.log | cut -c1-120 | tail -8
kill $APID 2>/dev/null; wait $APID 2>/dev/null
echo "--- C: tracer process after A is gone"; ( time timeout -s KILL 60 python tools/debug/ktrace_probe.py > /tmp/c.log 2>&1 ) 2>&1 | grep real; grep "probe" /tmp/c.log | cut -c1-100 | tail -3
