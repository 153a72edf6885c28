for i in 1 2 3 4 5 6 7 8; do
  PYTHONUNBUFFERED=1 timeout 600 python -m pytest tests/test_combine_gpu.py tests/test_concurrent_gpu.py tests/test_multidev_gpu.py "tests/test_bsw_gpu.py::test_concurrent_host_threads" -m gpu -q -p no:cacheprovider --timeout 120 -x 2>&1 | tail -3
done
echo "== stress_concurrent"; timeout 400 python scripts/stress_concurrent.py 240 12 11 2>&1 | tail -5
echo "== refdrivers loop"
for i in 1 2 3 4 5 6; do timeout 120 python scripts/refdrivers_large.py bsw --pairs 300000 2>&1 | grep -c "SW region" ; done
