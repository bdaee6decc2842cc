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
out, vals = [], {}
for name in ('fetch1024', 'write1024'):
    for d in glob.glob(f'{O}/{name}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(d)):
            if 'k2b' in r['Kernel_Name']:
                kn = r['Kernel_Name'].split('(')[0]
                out.append((kn, r['Grid_Size'], r['Counter_Name'], r['Counter_Value']))
                if 'fit_world' in kn or ('lbs_mfma' in kn and int(r['Grid_Size']) > 100000):
                    vals.setdefault(('fit' if 'fit_world' in kn else 'lbs', r['Counter_Name']), []).append(float(r['Counter_Value']))
with open('profiles/r01_final_hbm_pmc_1024f.csv', 'w') as f:
    f.write('kernel,grid_size,counter,value_KiB\n')
    for o in out: f.write(','.join(o) + '\n')
avg = {k: sum(v) / len(v) for k, v in vals.items()}
print(avg)
t = {"_doc": "HBM bytes per launch at 1024 frames from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for wide coalesced reads). KiB -> bytes.",
     "fit_frames_1024": int((2 * avg[('fit', 'FETCH_SIZE')] + avg[('fit', 'WRITE_SIZE')]) * 1024),
     "lbs_frames_1024": int((2 * avg[('lbs', 'FETCH_SIZE')] + avg[('lbs', 'WRITE_SIZE')]) * 1024)}
json.dump(t, open('profiles/traffic_r01.json', 'w'), indent=1)
print(t)
for n in ('1024', '4096'):
    rows = list(csv.DictReader(open(f'profiles/r01_final_kernel_stats_{n}f.csv')))
    for r in rows[:4]: print(n, r['Name'][:60], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
    b = json.load(open(f'profiles/r01_final_bench_{n}.json')); print(b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['frac'], b['roofline_lbs']['avg_launch_ms'], b['roofline_lbs']['frac'], b.get('cpu_baseline', {}).get('value'))
    b = json.load(open(f'profiles/r01_final_bench_{n}_under_rocprof.json')); print('under rocprof', b['roofline']['avg_launch_ms'], b['roofline_lbs']['avg_launch_ms'])
