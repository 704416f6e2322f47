O=gpurun_out/r05j
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_reference.py -m gpu -q -p no:cacheprovider -k "bench or scale or ddp or launcher" 2>&1 | tail -40 > $O/pytest.log
tail -6 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05j/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "ms_per_step_iqr", "session", "single_view_in_flight_views_per_s", "forward_only_views_per_s", "deferred_check_views_per_s")})
print(json.dumps(d["roofline"])[:1500])
print(json.dumps(d["stages"]))
print(json.dumps(d["stage2_gradient_set"]), json.dumps(d["stage2_path"])[:300])
c = d["configs"] or {}
for k, v in c.items():
    print(k[:40], json.dumps(v.get("fwd_bwd")), json.dumps(v.get("stages_one_view_in_flight"))[:1200] if v.get("stages_one_view_in_flight") else "", v.get("gradient_records"))
fp = d["full_pipeline"] or {}
print({k: (v.get("stage2_iters_per_s") or v.get("views_per_s_within_sample")) for k, v in fp.items() if isinstance(v, dict) and v.get("measured_in_this_run")})
print(d["cpu_baseline"], d["hip_graph_replay"])
PY
