cd /root/repo
timeout 900 python -m pytest tests/test_gpu_merge.py tests/test_merge_golden.py -m gpu -x -q 2>&1 | tail -5
for cfg in "2000000 100" "3000000 72" "1600000 128"; do
  timeout 900 python tools/merge_check.py $cfg 5 2>&1 | grep -E "RESULT|spot"
  MIDAS_SNPS_LIBRARY=midas_amd/lib/libmidas_snps_hip_twopass.so timeout 900 python tools/merge_check.py $cfg 5 2>&1 | grep -E "RESULT" | sed 's/^/TWOPASS /'
done
