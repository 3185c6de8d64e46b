cd /root/repo
mkdir -p gpurun_out/reh8
for cfg in "cg:" "gmres:--solver gmres --precond ilu0" "c4:--solver bicgstab --precond mcsgs --format ell" "c5:--solver mixed"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  st="--steps 40 --warmup 10"; [ $name = c5 ] && st="--steps 6 --warmup 2"
  ( time timeout 900 python bench.py --gpus 8 --transport callback $flags $st > gpurun_out/reh8/$name.json 2> gpurun_out/reh8/$name.err ) 2>&1 | grep real
  echo "$name rc=$? $(cut -c1-300 gpurun_out/reh8/$name.json)"
done
