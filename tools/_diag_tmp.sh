run() { python tools/conv_probe.py $1 200 2>/dev/null; }
for v in pipe_w0 pipe_nomfma pipe_w0_nomfma; do
  echo "== variant '$v'"
  export DAM_LIB_PATH=tools/libdam_$v.so
  run layer3
  DAM_TILE=1x2 run layer3 | sed 's/^/ 1x2 /'
  run layer4
  run layer5
  run layer6
done
