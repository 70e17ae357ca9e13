"""Worker of the forced-collectives tests: ONE rank drives every torch.distributed call of the multi-GPU path against itself.

    python tests/_nccl_worker.py ROOT BACKEND OUTFILE [hang]        (BACKEND: nccl (= RCCL, GPU 0) | gloo (host, stub engine))

nccl: init_process_group('nccl', world_size=1, device_id=cuda:0) + barrier(device_ids=[0]) (distributed.init_collectives), then
  (1) gather_table: the device float64 all_gather_into_tensor, with and without the plan's row counts;
  (2) EqvExchange: a self-transfer through batch_isend_irecv on device tensors -- the payload is produced by kernels that are still
      queued when the send is issued, the receive buffer is poisoned with NaN first, the consumer runs without any host synchronisation;
  (3) run_plan with the real engine and self-transfers, against the same plan without transfers (bitwise), and against the same
      exchange staged through a gloo group of the same process (host payloads; bitwise);
  every section under the hang watchdog (distributed.watchdog: the process ends with status 1 if a call does not return).
gloo: the same control flow on the host with the stub engine (CPU test of the self-transfer branch).
hang: arms the watchdog around a sleep -- the process must die with status 1 (tests the watchdog itself)."""
import os
import sys
import time

import numpy as np
import torch


def main():
    root, backend, outfile = sys.argv[1], sys.argv[2], sys.argv[3]
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    from roreg_amd import distributed as D
    if len(sys.argv) > 4 and sys.argv[4] == 'hang':
        with D.watchdog(1.0, 'a call that never returns'):
            time.sleep(30)
        print('the watchdog did not fire'); sys.exit(0)
    os.environ['ROREG_FORCE_COLLECTIVES'] = '1'
    on_gpu = backend == 'nccl'
    if on_gpu:
        assert torch.cuda.is_available()
        torch.cuda.set_device(0)
    dist = D.init_collectives(backend, 0, 1, 0 if on_gpu else None, timeout_s=60.0)
    assert dist.get_backend() == backend and dist.get_world_size() == 1
    out = {}
    dev = 'cuda' if on_gpu else 'cpu'

    # (1) the result-table collective
    rng = np.random.default_rng(5)
    tab = rng.standard_normal((37, D.ROW)); tab[3, 7] = np.nan; tab[5, 2] = 2.0 ** 52 + 1
    with D.watchdog(60.0, 'gather_table'):
        a = D.gather_table(tab, device=dev, counts=[37])
        b = D.gather_table(tab, device=dev)                       # sizes exchanged first (all_gather of the counts)
        e = D.gather_table(np.zeros((0, D.ROW)), device=dev, counts=[0])
    assert a.tobytes() == tab.tobytes() and b.tobytes() == tab.tobytes() and e.shape == (0, D.ROW)
    out['table'] = a

    if on_gpu:
        # (2) raw exchange: stream order.  `src` is written by a chain of kernels that is still queued when isend is issued (no host sync
        # anywhere); the receive buffer holds NaN until the transfer lands; the consumer is enqueued right behind wait().
        n = 5000 * 32 * 60
        base = torch.arange(n, dtype=torch.float32, device='cuda').reshape(5000, 32, 60)
        with D.watchdog(60.0, 'EqvExchange self-transfer'):
            src = base.clone()
            for _ in range(200):                                   # ~200 passes over 38.4 MB: milliseconds of queued work behind which the send must wait
                src = src * 1.0009765625 + 1.0
            want = base.double()
            for _ in range(200):
                want = (want * 1.0009765625).float().double() + 1.0
                want = want.float().double()
            ex = D.EqvExchange(0)
            assert ex.on_device
            recv = torch.full((5000, 32, 60), float('nan'), device='cuda')
            ex.start([('s', 0, 0, 0)], lambda s, i: src, lambda s, i: recv)
            got = ex.wait()[('s', 0)]
            consumer = got * 1.0                                   # enqueued on the current stream, no host sync before it
            torch.cuda.synchronize()
        assert got.data_ptr() == recv.data_ptr()
        assert torch.equal(consumer, src) and not torch.isnan(consumer).any()
        assert torch.equal(src.double(), want)
        assert ex.bytes_sent == ex.bytes_received == n * 4
        out['raw_exchange_bytes'] = np.int64(ex.bytes_sent)

    # (3) run_plan through the exchange branch
    from roreg_amd import synth
    from roreg_amd.parses.parses_test import default_config
    if on_gpu:
        from roreg_amd.engine import RegistrationEngine
        from roreg_amd.network import name2network
        cfg = default_config(keynum=96, ET='yohoo', max_iter=200)
        gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
        et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
        engine = RegistrationEngine(cfg, gf, et)
        n_kpts = 128
    else:
        from _dist_worker import StubEngine

        class Engine(StubEngine):
            def extract_many(self, feats, keys):
                return [type('C', (), {'eqv': torch.from_numpy(np.asarray(f, np.float32)) * 2.0})() for f in feats]

            def alloc_eqv(self, before):
                return torch.full(tuple(before.shape), float('nan'))

            def cloud_from_eqv(self, before, eqv, keys):
                assert torch.equal(eqv, torch.from_numpy(np.asarray(before, np.float32)) * 2.0)
                return type('C', (), {'eqv': eqv})()
        engine = Engine()
        cfg = default_config(keynum=16, ET='yohoo')
        n_kpts = 16
    scenes = {'a': synth.make_scene(5, n_clouds=6, n_kpts=n_kpts, overlap=0.6, name='synth/a'),
              'b': synth.make_scene(9, n_clouds=5, n_kpts=n_kpts, overlap=0.6, name='synth/b')}
    lists = {s: scenes[s].pair_ids for s in scenes}
    pieces = [('a', 0, len(lists['a'])), ('b', 0, len(lists['b']))]

    def scene_inputs(s):
        ds = scenes[s]
        return ({int(i): ds.feats[q] for q, i in enumerate(ds.pc_ids)}, {int(i): ds.get_kps(i) for i in ds.pc_ids}, ds.pair_ids,
                [1000 + q for q in range(len(ds.pair_ids))])

    transfers = D.self_transfers(pieces, lists, rank=0, n_clouds=3)
    assert len(transfers) == 3 and all(t[0] == 'b' and t[2] == t[3] == 0 for t in transfers)

    def rows_of(done):
        return np.stack([np.concatenate([[float(r.id0), float(r.id1), r.n_match, r.recalltime], np.asarray(r.trans, np.float64).reshape(-1)])
                         for _, _, _, res in done for r in res])

    kw = dict(keynum=cfg.keynum, max_iter=cfg.max_iter) if on_gpu else {}
    with D.watchdog(120.0, 'run_plan'):
        plain = rows_of(D.run_plan(engine, pieces, scene_inputs, [], 0, **kw))
        st = {}
        forced = rows_of(D.run_plan(engine, pieces, scene_inputs, transfers, 0, stats=st, **kw))
        if on_gpu:
            torch.cuda.synchronize()
    assert plain.tobytes() == forced.tobytes(), 'the exchange branch changed a result'
    assert st['eqv_bytes_sent'] == st['eqv_bytes_received'] == 3 * n_kpts * 32 * 60 * 4
    out['rows'] = forced
    if on_gpu:
        g = dist.new_group(backend='gloo')
        with D.watchdog(120.0, 'run_plan with a gloo-staged exchange'):
            staged = rows_of(D.run_plan(engine, pieces, scene_inputs, transfers, 0, exchange=D.EqvExchange(0, device_payloads=False, group=g), **kw))
            tab_gloo = D.gather_table(tab, device='cpu', counts=[37], group=g)
        assert staged.tobytes() == forced.tobytes() and tab_gloo.tobytes() == a.tobytes()
    with D.watchdog(60.0, 'final barrier'):
        dist.barrier(device_ids=[0]) if on_gpu else dist.barrier()
    np.savez(outfile, **out)
    dist.destroy_process_group()
    print('ok')


if __name__ == '__main__':
    main()
