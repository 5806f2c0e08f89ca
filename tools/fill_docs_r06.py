import json, re, sys
R='/root/repo/'
d=json.load(open(R+'profiles/r06/bench_shape_default.json'))
e=d['extra']; rf=d['roofline']
def ms(x): return "%.3f" % x
tests=open(R+'profiles/r06/pytest_gpu_tail.txt').read()
m=re.search(r'(\d+) passed, (\d+) skipped', tests)
T = "%s passed, %s skipped" % (m.group(1), m.group(2)) if m else "?"
bs=e['bait_sweep']
def leg(k):
    v=bs[k]; r=v['roofline']
    if r['bound'] == 'l2_gather':
        wp = r.get('whole_pass_frac') or r['lookups_per_launch'] / (v['ms_per_step'] * 1e-3) / 1e9 / r['peak']
        roof = "samples/s over the gather roof %.2f a pass, %.2f a launch" % (wp, r['frac'])
    else:
        roof = {"hbm": "HBM %.2f" % r['frac'], "valu": "issue %.2f" % (r['frac'] or 0)}[r['bound']]
    return "%s (%.3f; %s)" % (ms(v['ms_per_step']), v['whole_pass_frac_of_hbm_peak'], roof)
bait = " / ".join(leg(k) for k in ("33000","100000","350000","1000000","8500000"))
bait_short = " / ".join("%s (%.3f)" % (ms(bs[k]['ms_per_step']), bs[k]['whole_pass_frac_of_hbm_peak']) for k in ("33000","100000","350000","1000000","8500000"))
ks=e['k_sweep']
th=e['threshold_sweep']
f=e['e2e_files']; c4=f['configs4_se_gz']
warm=sorted(c4['seconds_each'][1:])
pl4=f['configs4_se_plain']; pl1=f['configs1_pe_plain']
fv=e['filter_v2']
dec=open(R+'profiles/r06/h_gzdev_check_under_rocprof.log').read()
mdec=re.findall(r'kernel [\d.]+ ms: ([\d.]+) GB/s of text', dec)
sub = {
 'R6_TESTS': T,
 'R6_VALUE': "**%.3f × 10¹¹ reads/s**, %.4f ms" % (d['value']/1e11, d['ms_per_step']),
 'R6_ROOF': "%.3f / %.3f / %.3f" % (rf['frac'], rf['kernel_alone_frac'], rf['whole_pass_frac']),
 'R6_TRAFFIC': "%.4f GB a launch against %.4f GB algorithmic (%.2f)" % (rf['traffic']/1e9, rf['algorithmic_bytes_per_launch']/1e9, rf['traffic']/rf['algorithmic_bytes_per_launch']),
 'R6_CPU': "%.2f M reads/s, C oracle, %d threads (16-CPU quota), all 33.3 M bits equal: %s" % (d['cpu_baseline']['value']/1e6, d['cpu_baseline']['cores'], e.get('sample_bits_match_oracle')),
 'R6_BAIT_SHORT': bait_short,
 'R6_BAIT': "**" + bait + "**; every leg's 1.5 M-read window equal to the oracle: %s" % all(bs[k].get('window_bits_match_oracle') for k in bs),
 'R6_K': "%.3f (%.3f of vector issue) / **%.3f** (a launch: %.3f -- with three buffer sets two or three screens are in flight)" % (ks['21']['whole_pass_frac_of_hbm_peak'], ks['21']['roofline'].get('frac') or 0, ks['41']['whole_pass_frac_of_hbm_peak'], ks['41']['roofline'].get('frac') or 0),
 'R6_T': "%s / %s / %s–%s ms; exhaustive %.2f ms = **%.3f of the vector issue rate**" % (ms(th['2']['ms_per_step']), ms(th['7']['ms_per_step']), ms(min(th[k]['with_hit_counts']['ms_per_step'] for k in '127')), ms(max(th[k]['with_hit_counts']['ms_per_step'] for k in '127')), th['exhaustive']['ms_per_step'], th['exhaustive']['roofline']['frac']),
 'R6_RAGGED': "%s ms a pass = **%.2f ×** uniform (%.3f of HBM), window equal to the oracle: %s" % (ms(e['ragged']['ms_per_step']), e['ragged']['ms_per_step_over_uniform'], e['ragged']['whole_pass_frac_of_hbm_peak'], e['ragged']['window_bits_match_oracle']),
 'R6_REAL': "%s ms a pass = %.2f × the iid set, **%.3f of HBM**, %.3f work items a read, window equal to the oracle: %s" % (ms(e['realistic']['ms_per_step']), e['realistic']['ms_per_step_over_iid'], e['realistic']['whole_pass_frac_of_hbm_peak'], e['realistic']['work_items_per_read'], e['realistic']['window_bits_match_oracle']),
 'R6_C4COLD': "%.2f s / %.2f–%.2f s" % (c4['first_call_seconds'], min(c4['cli_cold']['seconds_each']), max(c4['cli_cold']['seconds_each'])),
 'R6_C4': "**%.3f s = %.1f M reads/s**, %.3f of the PCIe roof; warm calls %s (max / min %.2f); inflate kernels %.1f GB/s of text in the pipeline; %.1f GB in use" % (c4['seconds'], c4['reads_per_s']/1e6, c4['roofline']['frac'], ", ".join("%.3f" % x for x in c4['seconds_each'][1:]), warm[-1]/warm[0], c4['inflate_kernels']['text_GB_per_s'], c4['device_memory_in_use_peak_GB']),
 'R6_PLAIN': "**%.3f s** (median %.3f, %.2f of the PCIe roof; `library_default` %.3f / %.3f) / %.3f s (median %.3f; %.3f / %.3f)" % (pl4['device_path']['seconds'], pl4['device_path']['seconds_median'], pl4['roofline']['frac'], pl4['library_default']['seconds'], pl4['library_default']['seconds_median'], pl1['device_path']['seconds'], pl1['device_path']['seconds_median'], pl1['library_default']['seconds'], pl1['library_default']['seconds_median']),
 'R6_FV2': "%.3f / %.3f s; **%.1f GB**; %.3f / %.3f s" % (fv['library_call']['device']['seconds'], fv['library_call']['host']['seconds'], fv['library_call']['device']['device_memory_in_use_peak_GB'], fv['cli_process_start_to_exit']['device']['seconds'], fv['cli_process_start_to_exit']['host']['seconds']),
 'R6_DEC': "%s GB/s" % (mdec[-1] if mdec else "?"),
}
for fn in ('DESIGN.md', 'README.md', 'HISTORY.md'):
    s=open(R+fn).read()
    for k in sorted(sub, key=len, reverse=True):
        s=s.replace(k, sub[k])
    left=re.findall(r'R6_[A-Z0-9_]+', s)
    if left: print(fn, 'unfilled', set(left))
    open(R+fn,'w').write(s)
print(json.dumps(sub, indent=1, ensure_ascii=False))
