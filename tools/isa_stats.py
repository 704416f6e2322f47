"""Static instruction mix of the kernels of one HIP unit (cross-compiled for gfx950, no GPU needed).
python tools/isa_stats.py gsr_composite_tiles.hip [kernel-substring] [--dump FILE] [extra hipcc flags...]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
args = sys.argv[2:]
dump = None
if "--dump" in args:
    i = args.index("--dump"); dump = args[i + 1]; del args[i:i + 2]
want = args[0] if args and not args[0].startswith("-") else ""
extra = [a for a in args if a.startswith("-")]
path = src if os.path.exists(src) else os.path.join(ROOT, "gps-gaussian_amd", "csrc", src)
unit = os.path.basename(path)
flags = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fhip-fp32-correctly-rounded-divide-sqrt"]
if unit in ("gsr_preprocess.hip",): flags += ["-ffp-contract=off", "-fno-slp-vectorize"]
if unit in ("corr_sampler.hip", "pack_views.hip", "unproject.hip", "gsr_binning.hip"): flags += ["-ffp-contract=off"]
if unit == "gsr_binning.hip": flags += ["-mllvm", "-simplifycfg-sink-common=false"]
if unit.startswith("gsr_composite"): flags += ["-fno-slp-vectorize"]
out = "/tmp/isa_%s.s" % unit
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *flags, *extra, "-S", "--cuda-device-only", "-o", out, path], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
meta = {}
blk = {}
for l in lines:  # amdhsa.kernels metadata: one YAML list item per kernel ("  - .agpr_count: ..."), .name somewhere inside it
    if re.match(r"\s+- \.", l):
        if "name" in blk: meta[blk["name"]] = blk
        blk = {}
    m = re.match(r"\s+(?:- )?\.name:\s+(\S+)", l)
    if m and "name" not in blk and not l.strip().startswith("- .name") or (m and l.strip().startswith(".name:") and "name" not in blk):
        blk["name"] = m.group(1)
    m = re.match(r"\s+(?:- )?\.(vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\d+)", l)
    if m: blk[m.group(1)] = int(m.group(2))
if "name" in blk: meta[blk["name"]] = blk
for v in meta.values(): v.pop("name", None)
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if m and ("k_" in m.group(1)):
        name = m.group(1)
        j = i
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"): j += 1  # (a kernel with early exits has several s_endpgm)
        body = lines[i:j + 1]
        if want in name:
            ops = [re.match(r"\s+([a-z_0-9]+)", l).group(1) for l in body if re.match(r"\s+[vsd]_|\s+(global|buffer|flat|ds)_", l)]
            c = collections.Counter(ops)
            cls = collections.Counter()
            for o, n in c.items():
                k = "mfma" if "mfma" in o else "valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "lds" if o.startswith("ds_") else "vmem"
                cls[k] += n
            short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:40]
            print(short, dict(cls), meta.get(name, {}))
            print("   top:", ", ".join("%s %d" % kv for kv in c.most_common(14)))
            if dump: open(dump, "w").write("\n".join(body))
        i = j
    i += 1
