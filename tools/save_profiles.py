"""Copy the measurement bundle of tools/profile_round.sh from gpurun_out/prof_<TAG>/ into profiles/
(files the judge reads) and refresh profiles/traffic_r01.json.  usage: python tools/save_profiles.py TAG"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
O = f'gpurun_out/prof_{tag}'
for f in glob.glob('profiles/r01_final_*'): os.remove(f)
shutil.copy(glob.glob(O + '/stats1024/*/*kernel_stats.csv')[0], 'profiles/r01_final_kernel_stats_1024f.csv')
shutil.copy(glob.glob(O + '/stats4096/*/*kernel_stats.csv')[0], 'profiles/r01_final_kernel_stats_4096f.csv')
for n in ('bench_1024', 'bench_4096', 'bench_1024_under_rocprof', 'bench_4096_under_rocprof'):
    shutil.copy(f'{O}/{n}.json', f'profiles/r01_final_{n}.json')
t = {"_doc": "HBM bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for wide coalesced reads). KiB -> bytes."}
for frames in ('1024', '4096'):
    out, vals = [], {}
    for name in (f'fetch{frames}', f'write{frames}'):
        for d in glob.glob(f'{O}/{name}/*/*counter_collection.csv'):
            for r in csv.DictReader(open(d)):
                if 'k2b' in r['Kernel_Name']:
                    kn = r['Kernel_Name'].split('(')[0]
                    out.append((kn, r['Grid_Size'], r['Counter_Name'], r['Counter_Value']))
                    if 'fit_world' in kn or 'lbs_mfma' in kn:
                        vals.setdefault(('fit' if 'fit_world' in kn else 'lbs', r['Counter_Name']), []).append((int(r['Grid_Size']), float(r['Counter_Value'])))
    if not vals:
        continue
    with open(f'profiles/r01_final_hbm_pmc_{frames}f.csv', 'w') as f:
        f.write('kernel,grid_size,counter,value_KiB\n')
        for o in out: f.write(','.join(o) + '\n')
    # full-mesh launches only: the LBS kernel also runs on the 21 vertex-selected joints with a small grid
    avg = {}
    for k, v in vals.items():
        gmax = max(g for g, _ in v)
        sel = [x for g, x in v if g == gmax]
        avg[k] = sum(sel) / len(sel)
    print(frames, avg)
    t[f"fit_frames_{frames}"] = int((2 * avg[('fit', 'FETCH_SIZE')] + avg[('fit', 'WRITE_SIZE')]) * 1024)
    t[f"lbs_frames_{frames}"] = int((2 * avg[('lbs', 'FETCH_SIZE')] + avg[('lbs', 'WRITE_SIZE')]) * 1024)
json.dump(t, open('profiles/traffic_r01.json', 'w'), indent=1)
print(t)
for n in ('1024', '4096'):
    rows = list(csv.DictReader(open(f'profiles/r01_final_kernel_stats_{n}f.csv')))
    for r in rows[:4]: print(n, r['Name'][:60], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
    b = json.load(open(f'profiles/r01_final_bench_{n}.json')); print(b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['frac'], b['roofline_lbs']['avg_launch_ms'], b['roofline_lbs']['frac'], b.get('cpu_baseline', {}).get('value'))
    b = json.load(open(f'profiles/r01_final_bench_{n}_under_rocprof.json')); print('under rocprof', b['roofline']['avg_launch_ms'], b['roofline_lbs']['avg_launch_ms'])
