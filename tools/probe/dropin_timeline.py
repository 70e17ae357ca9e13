"""Workload for a GPU timeline of the evaluator's engine route (tools/probe/dropin_timeline.sh runs it under rocprofv3 --kernel-trace
--memory-copy-trace): bench.py's dropin scene on disk, `process_scene` three times (the third is the one to read), then the engine alone on
resident inputs twice -- 0.5 s of sleep between the runs so the report can cut the trace into windows."""
import os, sys, time, tempfile, shutil
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from roreg_amd import synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
from roreg_amd.test import _cache
from roreg_amd.test.evaluator import yoho_evaluator

n_clouds, n_pairs, kpts = 60, 449, 5000
torch.manual_seed(0)
ET = 'yohoc' if '--yohoc' in sys.argv else 'yohoo'
RD = '--rd-rm' in sys.argv
cfg0 = default_config(keynum=kpts, max_iter=1000, ET=ET, RD=RD, RM=RD)
gf_sd = synth.seeded_state_dict(name2network['GF_test'](cfg0), 101)
et_sd = synth.seeded_state_dict(name2network['ET_test'](cfg0), 202)
root = tempfile.mkdtemp(prefix='roreg_dropin_tl_')
try:
    feats, keys, poses = synth.make_scene_device(777, n_clouds, kpts, 0.6)
    pairs = synth.scene_pair_list(n_clouds, n_pairs, 901, locality=8.0)
    ds = synth.SynthScene('synth/kitchen', [k.cpu().numpy() for k in keys], None, poses, pairs)
    inputs = f'{root}/inputs/FCGF_Input_Group_feature'
    os.makedirs(inputs)
    for i, f in enumerate(feats):
        np.save(f'{inputs}/{i}.npy', f.cpu().numpy())
    nets = [('GF', gf_sd), ('ET', et_sd)]
    if RD:
        import bench
        rd, rm, _ = bench.rd_rm_nets(cfg0)
        nets += [('RD', rd.state_dict()), ('RM', rm.state_dict())]
    for kind, sd in nets:
        os.makedirs(f'{root}/ckpt/{kind}')
        torch.save({'best_para': 0, 'network_state_dict': sd}, f'{root}/ckpt/{kind}/model_best.pth')
    cfg = default_config(output_cache_fn=f'{root}/cache_0', model_fn=f'{root}/ckpt', base_dir=root, SO3_related_files=None, keynum=kpts, max_iter=1000, ET=ET, RD=RD, RM=RD)
    ev = yoho_evaluator(cfg)
    for rep in range(5):
        cache = f'{root}/cache_{rep}'
        cfg.output_cache_fn = cache
        os.makedirs(f'{cache}/{ds.name}')
        os.symlink(inputs, f'{cache}/{ds.name}/FCGF_Input_Group_feature')
        _cache.clear()
        np.random.seed(5)
        prof = None
        samples, stop = [], None
        if rep >= 1 and '--sample' in sys.argv:                  # where the launching thread sits, every 0.5 ms
            import threading, traceback
            main_id = threading.main_thread().ident
            stop = threading.Event()
            def sampler():
                while not stop.is_set():
                    fr = sys._current_frames().get(main_id)
                    stack = []
                    while fr is not None and len(stack) < 4:
                        stack.append(f'{os.path.basename(fr.f_code.co_filename)}:{fr.f_lineno}({fr.f_code.co_name})'); fr = fr.f_back
                    samples.append((time.perf_counter(), ' < '.join(stack)))
                    time.sleep(0.0005)
            threading.Thread(target=sampler, daemon=True).start()
        if rep == 4 and '--cprofile' in sys.argv:
            import cProfile
            prof = cProfile.Profile()
        torch.cuda.synchronize(); time.sleep(0.5); t0 = time.perf_counter()
        if prof: prof.enable()
        ev.process_scene(ds)
        if prof: prof.disable()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if stop is not None:
            stop.set()
            runs, cur = [], None
            for t, w in samples:
                if t < t0: continue
                if cur is not None and cur[2] == w: cur[1] = t
                else:
                    cur = [t, t, w]; runs.append(cur)
            print(f'   scene {rep}: where the launching thread stayed >= 4 ms (offset ms, duration ms, innermost frames):')
            for a_, b_, w in runs:
                if b_ - a_ >= 0.004:
                    print(f'      {1e3 * (a_ - t0):7.1f} {1e3 * (b_ - a_):6.1f}  {w}')
        if prof:
            import io, pstats
            for key in ('tottime', 'cumtime'):
                st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats(key).print_stats(32); print(st.getvalue()[:7000])
        print(f'engine route, scene {rep}: {dt:.4f} s = {n_pairs / dt:.1f} pairs/s  {getattr(ev, "last_scene_seconds", None)}', flush=True)
        tl = getattr(ev, 'last_scene_timeline', None)
        if tl:
            print('   host marks (ms from the start of process_scene): ' + ', '.join(f'{k} {1e3 * v:.1f}' for k, v in tl), flush=True)
        shutil.rmtree(cache, ignore_errors=True)
    eng = ev._engine()
    for rep in range(3):
        np.random.seed(5)
        torch.cuda.synchronize(); time.sleep(0.5); t0 = time.perf_counter()
        eng.run_scene(feats, keys, ds.pair_ids, all_local_transforms=(ET == 'yohoo'))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'engine alone, resident inputs, run {rep}: {dt:.4f} s = {n_pairs / dt:.1f} pairs/s', flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
