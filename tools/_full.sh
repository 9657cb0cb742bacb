cd /root/repo
cp gpurun_out/parity_report.jsonl /tmp/pr_before.jsonl 2>/dev/null
rm -f gpurun_out/parity_report.jsonl
ZEDO_MATH=f16x3 timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider -rs > gpurun_out/suite_f16x3.log 2>&1; echo "f16x3 rc=$?"; tail -n 25 gpurun_out/suite_f16x3.log | cut -c1-220
cp gpurun_out/parity_report.jsonl gpurun_out/parity_report_r05_f16x3.jsonl
