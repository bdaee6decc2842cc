"""Copy the measurement bundle of tools/profile_round.sh from gpurun_out/prof_<TAG>/ into profiles/ (files the judge
reads) and write profiles/traffic_<ROUND>.json.  usage: python tools/save_profiles.py TAG [ROUND=r03]"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
O = f'gpurun_out/prof_{tag}'
# gpurun merges a run's files into gpurun_out/ without removing older ones: keep only the newest run (files within 40 minutes
# of the newest file of the bundle), so that a rerun is never mixed with the one before it
_all = [os.path.join(r, f) for r, _, fs in os.walk(O) for f in fs]
_new = max(os.path.getmtime(f) for f in _all)
for f in _all:
    if os.path.getmtime(f) < _new - 2400: os.remove(f)
P = f'profiles/{rnd}_final_'
for f in glob.glob(P + '*'): os.remove(f)
for d, n in (('stats4096', 'kernel_stats_4096seq'), ('stats1024', 'kernel_stats_1024f'), ('statsx1024', 'kernel_stats_smplx_1024f')):
    shutil.copy(glob.glob(f'{O}/{d}/*/*kernel_stats.csv')[0], f'{P}{n}.csv')
for n in ('bench_default', 'bench_1024', 'bench_smplx_1024', 'bench_smplx_4096', 'bench_default_under_rocprof',
          'bench_1024_under_rocprof', 'bench_smplx_1024_under_rocprof'):
    shutil.copy(f'{O}/{n}.json', f'{P}{n}.json')
t = {"_doc": "HBM bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of the bench command); "
             "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for wide coalesced reads). KiB -> bytes. "
             "lbs = pose set-up + vertex kernel (the stream kernels for SMPL and SMPL-X; the joint gather is part of it since round 3); lbs_tile_only = the vertex kernel alone."}
for key, pre in (('4096', ''), ('1024', ''), ('x1024', 'smplx_')):
    out, vals = [], {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        for d in glob.glob(f'{O}/{c}_{key}/*/*counter_collection.csv'):
            for r in csv.DictReader(open(d)):
                if 'k2b' not in r['Kernel_Name']:
                    continue
                kn = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
                out.append((kn, r['Grid_Size'], r['Counter_Name'], r['Counter_Value']))
                kind = 'fit' if 'fit_' in kn else ('tile' if ('lbs_tile' in kn or 'lbs_stream' in kn) else ('pose' if 'pose_setup' in kn else ('gather' if 'gather' in kn else None)))
                if kind:
                    vals.setdefault((kind, r['Counter_Name']), []).append((int(r['Grid_Size']), float(r['Counter_Value'])))
    if not vals:
        continue
    frames = key.lstrip('x')
    with open(f'{P}hbm_pmc_{pre}{frames}f.csv', 'w') as f:
        f.write('kernel,grid_size,counter,value_KiB\n')
        for o in out: f.write(','.join(o) + '\n')
    avg = {}
    for k, v in vals.items():                  # the bench's launches only (largest grid of that kernel)
        gmax = max(g for g, _ in v)
        sel = [x for g, x in v if g == gmax]
        avg[k] = sum(sel) / len(sel)
    print(key, {f"{a}/{b}": round(x) for (a, b), x in avg.items()})
    g = lambda kind: 2 * avg.get((kind, 'FETCH_SIZE'), 0.0) + avg.get((kind, 'WRITE_SIZE'), 0.0)
    t[f"{pre}fit_frames_{frames}"] = int(g('fit') * 1024)
    t[f"{pre}lbs_frames_{frames}"] = int((g('tile') + g('pose') + g('gather')) * 1024)
    t[f"{pre}lbs_tile_only_frames_{frames}"] = int(g('tile') * 1024)
json.dump(t, open(f'profiles/traffic_{rnd}.json', 'w'), indent=1)
print(t)
for n in ('4096seq', '1024f', 'smplx_1024f'):
    rows = list(csv.DictReader(open(f'{P}kernel_stats_{n}.csv')))
    for r in rows[:5]: print(n, r['Name'][:70], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
for n in ('bench_default', 'bench_1024', 'bench_smplx_1024', 'bench_default_under_rocprof', 'bench_1024_under_rocprof'):
    b = json.load(open(f'{P}{n}.json'))
    print(n, b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['frac'], b['roofline_lbs']['avg_launch_ms'],
          b['roofline_lbs']['frac'], b.get('cpu_baseline', {}).get('value'))

# SQ counter passes of the fit kernel (tools/pmc_fit.sh, run by profile_round.sh): raw per-dispatch rows + derived fractions
sq = {}
for fr in ("4096", "1024"):
    rawf, sumf = f'gpurun_out/pmc_{tag}_{fr}.raw.csv', f'gpurun_out/pmc_{tag}_{fr}.summary.json'
    if not (os.path.exists(rawf) and os.path.exists(sumf)):
        continue
    shutil.copy(rawf, f'profiles/{rnd}_sq_fit_{fr}.csv')
    a = json.load(open(sumf))["per_launch_average"]
    b = json.load(open(f'{P}bench_default.json' if fr == "4096" else f'{P}bench_1024.json'))
    ms = b['roofline']['avg_launch_ms']
    sq[fr] = {
        "valu_active_frac": round(a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"], 4),
        "wait_any_frac": round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 4),
        "mfma_busy_frac": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * ms * 1e-3 * 2.4e9), 4),
        "insts_valu_per_launch": a["SQ_INSTS_VALU"], "insts_mfma_per_launch": a["SQ_INSTS_MFMA"], "insts_lds_per_launch": a["SQ_INSTS_LDS"],
        "waves_per_launch": a["SQ_WAVES"], "lds_idx_active_cycles": a["SQ_LDS_IDX_ACTIVE"], "lds_bank_conflict_cycles": a["SQ_LDS_BANK_CONFLICT"],
        "launch_ms_in_bench": ms,
        "sq_source": f"profiles/{rnd}_sq_fit_{fr}.csv (tools/pmc_fit.sh: five rocprofv3 --pmc passes, last eight dispatches of each)",
    }
if sq:
    json.dump(sq, open(f'profiles/{rnd}_sq_fit.json', 'w'), indent=1)
    print(sq)
