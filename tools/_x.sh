for s in 0 1 2 0 1 2; do
  if [ $s = 0 ]; then unset TXE_X_DWS; else export TXE_X_DWS=$s; fi
  echo "== splits $s"; python tools/kernel_times.py --steps 20 2>&1 | grep -E "^step|second stream|4, 160>|bwd_dot|fused_bwd"
done
