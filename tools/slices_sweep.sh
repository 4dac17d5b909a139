out=gpurun_out/r02_slices.txt; : > $out
for cfg in "20 5 0:39 52 64 80 104 160" "64 64 64:12 17 24 32 48" ; do
  steps=$(echo $cfg | cut -d' ' -f1); warm=$(echo $cfg | cut -d' ' -f2); rest=$(echo $cfg | cut -d: -f2)
  for s in $rest; do
    echo "## steps $steps slices $s" >> $out
    POCS_SLICES=$s python bench.py --steps $steps --warmup $warm --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.4g frac %.3f' % (d['value'], d['roofline']['frac']))" >> $out
  done
done
for s in 96 128 192 256; do echo "## batch 8 slices $s" >> $out; POCS_SLICES=$s python bench.py --batch 8 --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.4g frac %.3f' % (d['value'], d['roofline']['frac']))" >> $out; done
cat $out
