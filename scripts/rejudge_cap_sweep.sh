cd /root/repo
for v in 24 32 48 64 96 160; do
  echo "cap $v: $(RT_REJ_FIRST_PASS_NFEV=$v timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-config2 --no-other-mode --no-extra-configs --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['kernel_ms_per_step']['rejudge'],2))")"
done
