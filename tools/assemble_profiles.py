"""copies what tools/collect_profiles.sh left in gpurun_out/<tag> into profiles/<tag>_* and recomputes
<tag>_pmc_traffic.json: python tools/assemble_profiles.py [tag]"""
import csv, json, os, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
o = 'gpurun_out/' + tag
shutil.copy(o + '/headline/t_kernel_stats.csv', 'profiles/' + tag + '_kernel_stats_headline_loop.csv')
shutil.copy(o + '/default/t_kernel_stats.csv', 'profiles/' + tag + '_kernel_stats_bench_default.csv')
shutil.copy(o + '/pmc_FETCH_SIZE_summary.csv', 'profiles/' + tag + '_pmc_fetch_size_summary.csv')
shutil.copy(o + '/pmc_WRITE_SIZE_summary.csv', 'profiles/' + tag + '_pmc_write_size_summary.csv')
shutil.copy(o + '/bench.json', 'profiles/' + tag + '_bench.json')            # the full detail of the default run
shutil.copy(o + '/bench_line.json', 'profiles/' + tag + '_bench_line.json')  # the compact line the driver parses
rows = lambda f: list(csv.DictReader(open(f)))
F, W = rows(o + '/pmc_FETCH_SIZE_summary.csv'), rows(o + '/pmc_WRITE_SIZE_summary.csv')
pick = lambda R, pat, col: [(r['kernel'], int(r['grid_size']), int(r['launches']), float(r[col])) for r in R if re.search(pat, r['kernel'])]
b = json.load(open(o + '/bench.json'))
out = {}
rf, rw = pick(F, r'^k_tcg_run<', 'FETCH_SIZE_median'), pick(W, r'^k_tcg_run<', 'WRITE_SIZE_median')
if rf and rw:
    rf, rw = max(rf, key=lambda x: x[2]), max(rw, key=lambda x: x[2])
    out['k_tcg_run_bytes_per_launch'] = (2 * rf[3] + rw[3]) * 1024
    out['k_tcg_run_counters'] = {'FETCH_SIZE_KB_median': rf[3], 'WRITE_SIZE_KB_median': rw[3], 'launches': rf[2],
        'note': 'one launch = one tCG run (z0 + about 7 iterations): traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 as the guide prescribes for 16-byte streaming reads (the rows of the dense inverse, 32 MB once per launch, and the sc1 gathers of H delta, 20 MB per iteration over 250 workgroups, are 16-byte loads); the median over launches of different iteration counts'}
pf = max(pick(F, r'^k_fused_pc<', 'FETCH_SIZE_median'), key=lambda x: x[2])
pw = max(pick(W, r'^k_fused_pc<', 'WRITE_SIZE_median'), key=lambda x: x[2])
out['k_fused_pc_bytes_per_launch'] = (2 * pf[3] + pw[3]) * 1024
out['k_fused_pc_counters'] = {'FETCH_SIZE_KB_median': pf[3], 'WRITE_SIZE_KB_median': pw[3], 'launches': pf[2],
    'note': 'traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts half of a 16-B/lane streaming read (MI355X_MICROARCH.md, HBM); bytes the kernel streams once 32.56 MB (dense inverse 32 MB + 7 r k vectors); the per-workgroup re-reads of r_old / H delta are L2 hits and do not reach the fabric counters; tools/pmc_run.py, tools/pmc_summarize.py, round 2'}
sf, sw = pick(F, r'^k_spmm_bsr[2q]?<', 'FETCH_SIZE_median'), pick(W, r'^k_spmm_bsr[2q]?<', 'WRITE_SIZE_median')
out['k_spmm_bsr_lattice100k_counters'] = {'kernels': [x[0] for x in sf], 'FETCH_SIZE_KB': [x[3] for x in sf], 'WRITE_SIZE_KB': [x[3] for x in sw], 'launches': [x[2] for x in sf],
    'note': '8-B gathers and 16-B block-row loads mixed: (FETCH+WRITE)*1024 and (2*FETCH+WRITE)*1024 bracket the traffic; algorithmic (CSR convention) 124.1 MB'}
# the replay of the lattice agent inside bench.roofline: the grids of k_sp_mtile<1, 8> launched most often
allf = pick(F, r'^k_sp_mtile<1, 8', 'FETCH_SIZE_median')
ncalls = max(x[2] for x in allf) if allf else 0
lf = [x for x in allf if x[2] == ncalls]
lw = [x for x in pick(W, r'^k_sp_mtile<1, 8', 'WRITE_SIZE_median') if x[2] == ncalls]
fk, wk = sum(x[3] for x in lf), sum(x[3] for x in lw)
out['k_sp_mtile_lattice100k_agent'] = {'distinct_grids_with_%d_calls_each' % ncalls: len(lf),
    'FETCH_SIZE_KB_per_launch': [x[3] for x in lf], 'WRITE_SIZE_KB_per_launch': [x[3] for x in lw],
    'FETCH_SIZE_KB_per_application': fk, 'WRITE_SIZE_KB_per_application': wk,
    'bytes_per_application_fetch_plus_write': (fk + wk) * 1024, 'bytes_per_application_2fetch_plus_write': (2 * fk + wk) * 1024,
    'algorithmic_bytes_per_application': b['roofline_qapply']['precond_sparse_lattice100k_agent']['bytes_per_application'],
    'note': 'k_sp_mtile (4-row tiles on the fp64 matrix pipe, round 4): 8-B weight loads and 16-B vector pair loads: (FETCH+WRITE)*1024 and (2*FETCH+WRITE)*1024 bracket the traffic; 7 level launches per application; launches of equal grid size are merged by the summary, so fewer than 7 rows may appear'}
json.dump(out, open('profiles/' + tag + '_pmc_traffic.json', 'w'), indent=1)
if os.path.exists('gpurun_out/qapply_pmc/summary.txt'):
    shutil.copy('gpurun_out/qapply_pmc/summary.txt', 'profiles/' + tag + '_qapply_counters.txt')
if os.path.exists(o + '/bench_2ranks_on_one_gpu.json') and os.path.getsize(o + '/bench_2ranks_on_one_gpu.json') > 100:
    shutil.copy(o + '/bench_2ranks_on_one_gpu.json', 'profiles/' + tag + '_bench_2ranks_on_one_gpu_gloo.json')
q = b['roofline_qapply']
print('value', b['value'], 'sustained', b['sustained']['value'], 'c5', b['config5_lattice100k']['value'], 'tiers', b['config4_tiers']['tcg_iterations_per_s'])
print('qapply warm/cold', q['qapply_lattice100k']['avg_launch_us'], q['qapply_lattice100k_cold']['avg_launch_us'], q['qapply_lattice100k']['kernel'])
print('replay', q['precond_sparse_lattice100k_agent']['avg_application_us'], 'central', b['config5_central_certified']['seconds_to_certified_optimum'], b['config5_central_certified']['preconditioner']['application_us'])
print('certified ms', b['ms_to_certified_optimum']['total_ms'], 'c2', b['config2_sphere2500_single']['tcg_iterations_per_s'], 'speedup', b['config']['speedup_vs_cpu_port'], b['sustained']['speedup_vs_cpu_port'])
print('pmc', json.dumps(out['k_spmm_bsr_lattice100k_counters']))
