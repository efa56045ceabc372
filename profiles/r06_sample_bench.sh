mkdir -p gpurun_out/r06
TAG=${1:-i}
python bench_denoise.py --mode sample --batch 64 --horizon 16 > gpurun_out/r06/${TAG}_sampling_cfg3.json 2> gpurun_out/r06/${TAG}_sampling_cfg3.err
python bench_denoise.py --mode sample --batch 24 --horizon 50 > gpurun_out/r06/${TAG}_sampling_L50.json 2> gpurun_out/r06/${TAG}_sampling_L50.err
A3D_DN_PERSIST=0 python bench_denoise.py --mode sample --batch 64 --horizon 16 > gpurun_out/r06/${TAG}_sampling_cfg3_perphase.json 2>/dev/null
for f in ${TAG}_sampling_cfg3 ${TAG}_sampling_L50 ${TAG}_sampling_cfg3_perphase; do python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r06/$f.json").read().strip().splitlines()[-1])
    print("$f", {k: d[k] for k in d if k in ("value","ms_per_denoise_step","unit","sampler")}, d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("ms"))
except Exception as e:
    print("$f", "ERR", e)
P
done
tail -2 gpurun_out/r06/${TAG}_sampling_cfg3.err
