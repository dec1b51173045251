"""on-device build vs CPU build at N x 100d: time and recall"""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import hnsw_rs_amd as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d, m = int(os.environ.get('DIM', '100')), 16
recipe = int(os.environ.get('RECIPE', '0'))
thr = min(32, os.cpu_count())
vs = H.synth_rows(recipe, 0x5EED0001, 0, N, d, thr); qs = H.synth_rows(recipe, 0x5EED0002, 0, 1024, d, 8)
mode = int(os.environ.get('MODE', '2'))
dev = H.HNSW.new(m, 32, d, kind); dev.set_option('gpu_build', mode)
if os.environ.get('BATCH'):  # "max:div", e.g. 32768:8 (bench.py --config 4)
    bmax, bdiv = (int(x) for x in os.environ['BATCH'].split(':'))
    dev.set_option('gpu_build_batch_max', bmax); dev.set_option('gpu_build_batch_div', bdiv)
t = time.time(); dev.insert_bulk_device(vs, thr, True); t_dev = time.time() - t
print('device build %.1fs layers=%d compliant=%s' % (t_dev, dev.nb_layers(), dev.assert_param_compliance()), flush=True)
g = lambda k: dev.stat('build_' + k) / 1e6
print('build clocks: insert kernel %.2f s (HIP events), insert phase %.2f s, sort + connect + remove phase %.2f s of which the '
      'connect + drop kernels %.2f s (HIP events); batch max %s' % (g('insert_kernel_us'), g('insert_phase_us'), g('connect_us'),
                                                                    g('connect_kernel_us'), os.environ.get('BATCH', 'default')), flush=True)
truth, _ = dev.brute_force(qs, 10)
for ef in (64, 96, 256):
    ids, _, _, st = dev.search_batch(qs, 10, ef)
    print('device-built ef=%d recall %.4f n_dist %.0f' % (ef, sum(len(set(a) & set(b)) for a, b in zip(ids.tolist(), truth.tolist())) / 10240, st[:, 0].mean()), flush=True)
if len(sys.argv) > 3:
    t = time.time(); cpu = H.HNSW.new(m, 32, d, kind).insert_bulk(vs, thr, False); print('cpu build %.1fs' % (time.time() - t), flush=True)
    for ef in (64, 96, 256):
        ids, _, _, st = cpu.search_batch(qs, 10, ef)
        print('cpu-built ef=%d recall %.4f n_dist %.0f' % (ef, sum(len(set(a) & set(b)) for a, b in zip(ids.tolist(), truth.tolist())) / 10240, st[:, 0].mean()), flush=True)
