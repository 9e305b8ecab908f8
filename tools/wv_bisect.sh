export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
f=0; for i in $(seq 1 30); do CANVAS_WV_LONG=256 CANVAS_WV_TRACE=1 python -m pytest tests/test_wavelets_gpu.py -q -m gpu -x -s -k "closed_form" > /tmp/t$i.log 2>&1; if grep -q failed /tmp/t$i.log; then f=$((f+1)); fi; grep "was complete before" /tmp/t$i.log | head -2; done; echo "$f of 30 failed"
python -m pytest tests/test_wavelets_gpu.py -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_wavelets_gpu.py -q -m gpu 2>&1 | tail -2
