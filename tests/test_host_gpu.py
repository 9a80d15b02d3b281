"""GPU: the drop-in API end to end -- Dataset items, ModelTrainer.fit (against the trajectory, stdout and checkpoint
names recorded from the reference's own ModelTrainer), the fused Adam, full-song inference."""
import json
import os

import numpy as np
import pytest
import torch

from _inputs import model_input
from oracle import features_ref, inference_ref, models_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dam(dam_lib):
    import deep_audio_mixer_amd as pkg
    return pkg


def test_dataset_items_match_oracle(dam):
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    from _inputs import feature_error
    rng = np.random.default_rng(4)
    sr = 16000
    songs = {'s%d' % i: {t: 0.1 * rng.standard_normal((int(d * sr), 2)) for t in ('bass', 'drums', 'vocals', 'other', 'mix')}
             for i, d in enumerate([3.2, 2.1])}
    d = MultitrackAudioDataset.from_arrays(songs, chunk_length=1, sr=sr)
    assert len(d) == 5
    loader = torch.utils.data.DataLoader(d, batch_size=2, shuffle=False, num_workers=0)
    batches = list(loader)
    assert batches[0][0].shape == (2, 4, 1025, 16) and batches[0][1].shape == (2, 1025, 16) and batches[0][0].is_cuda
    assert batches[0][0].dtype == torch.float32
    for index in (0, 4):
        x, gt = d[index]
        song_i, chunk_i = d._calculate_song_index(index)
        name = d.songlist[song_i]
        pcm = np.stack([songs[name][t][chunk_i * sr:(chunk_i + 1) * sr] for t in d.get_tracklist()])
        want_x, want_gt = features_ref.clip_features(pcm)
        for s in range(4):
            rel, db = feature_error(x[s].cpu().numpy(), want_x[s])
            assert rel <= 2e-6 and db <= 2e-3
        rel, db = feature_error(gt.cpu().numpy(), want_gt)
        assert rel <= 2e-6 and db <= 2e-3
    f = d.compute_features(songs['s0']['bass'][:sr].mean(1))
    assert f.shape == (1025, 16)


@pytest.mark.parametrize('opt_kind', ['own', 'torch'])
def test_trainer_matches_reference_run(dam, golden_dir, tmp_path, capsys, monkeypatch, opt_kind):
    """tests/golden/trainer.json was recorded from the reference's ModelTrainer.fit (CPU, torch Adam) on the same
    seeded batches and the same parameter fill: same stdout format, same checkpoint names, same loss trajectory.
    opt_kind 'torch': the notebook cell as written (training.ipynb cell 11) -- a plain torch.optim.Adam is handed to
    ModelTrainer, which adopts it into the fused launch: the captured path is taken and the caller's optimizer object
    stays current (moments, step counts, shared hyper-parameter dict)."""
    from deep_audio_mixer_amd.model_trainer import ModelTrainer
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    g = json.load(open(os.path.join(golden_dir, 'trainer.json')))
    ref = models_ref.closed_form_fill(models_ref.RefResNet18())
    model = ResNet18()
    model.load_state_dict(ref.state_dict())
    model = model.cuda()
    batches = [tuple(torch.from_numpy(a) for a in model_input(*g['shape'], seed=s)) for s in g['seeds']]
    train, val = batches[:2], batches[2:]
    monkeypatch.chdir(tmp_path)
    os.mkdir('weights')
    if opt_kind == 'own':
        opt = Adam(model.parameters(), lr=g['lr'], weight_decay=1e-5)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=g['lr'], weight_decay=1e-5)
    trainer = ModelTrainer(model, torch.nn.MSELoss(), opt, torch.device('cuda'), model_name='resnet')
    tl, vl = trainer.fit(train, val, g['start_epoch'], g['num_epochs'])
    n_steps = g['num_epochs'] * len(train)
    assert trainer.graph_steps == n_steps - trainer.EAGER_BATCHES and trainer.eager_steps == trainer.EAGER_BATCHES
    if opt_kind == 'torch':
        assert trainer._adopted_from is opt and isinstance(trainer.optimizer, Adam)
        assert trainer.optimizer.param_groups[0] is opt.param_groups[0]        # an LR scheduler on `opt` reaches the launch
        sd = opt.state_dict()                                                  # the caller's object is current
        assert len(sd['state']) == len(list(model.parameters()))
        assert all(float(st['step']) == n_steps for st in sd['state'].values())
        p0 = next(model.parameters())
        assert torch.equal(opt.state[p0]['exp_avg'], trainer.optimizer.state_dict()['state'][0]['exp_avg'])
        assert float(opt.state[p0]['exp_avg'].abs().max()) > 0
    out = capsys.readouterr().out.splitlines()
    assert len(out) == len(g['stdout'])
    for got, want in zip(out, g['stdout']):
        assert got.split(':')[0] == want.split(':')[0] and got.startswith(want[:10])     # same text, numbers checked below
    files = sorted(os.listdir('weights'))
    assert [f[:22] for f in files] == [f[:22] for f in g['files']]                        # mixmodel_resnet_1s_000N_
    # trajectory over 4 Adam steps (lr 1e-4).  Adam's first updates are ~lr*sign(g): entries whose gradient is
    # rounding-level flip sign between any two fp32 implementations, so the trajectory is compared at 1e-2
    np.testing.assert_allclose(tl, g['train_loss'], rtol=1e-2)
    np.testing.assert_allclose(vl[0], g['val_loss'][0], rtol=1e-2)
    np.testing.assert_allclose(vl[1], g['val_loss'][1], rtol=3e-2)      # after all 4 sign-like Adam updates
    sd = torch.load(os.path.join('weights', files[-1]))
    assert set(sd.keys()) == set(ref.state_dict().keys())
    np.testing.assert_allclose(sd['bn1.running_mean'].cpu().numpy(), g['bn1_running_mean'], rtol=0, atol=5e-3)
    # the weights MOVED as the reference's did: 4 Adam steps at lr 1e-4 change an entry by at most 4e-4, so the check is
    # on the delta (after - before), to within one step, with the right sign wherever the reference moved by > 1.5 steps
    before = ref.state_dict()['conv1.weight'].flatten()[:8].numpy().astype(np.float64)
    d_got = sd['conv1.weight'].flatten()[:8].cpu().numpy().astype(np.float64) - before
    d_want = np.asarray(g['conv1_weight_head']) - before
    assert np.abs(d_want).max() > 2 * g['lr']                      # the golden run did move these entries
    np.testing.assert_allclose(d_got, d_want, rtol=0, atol=g['lr'])
    big = np.abs(d_want) > 1.5 * g['lr']
    assert big.any() and np.array_equal(np.sign(d_got[big]), np.sign(d_want[big]))


def test_fused_adam_matches_torch_adam(dam):
    from deep_audio_mixer_amd.optim import Adam
    torch.manual_seed(0)
    ps = [torch.randn(257, 33, device='cuda'), torch.randn(1000, device='cuda'), torch.randn(3, 3, 3, 3, device='cuda')]
    a = [torch.nn.Parameter(p.clone()) for p in ps]
    b = [torch.nn.Parameter(p.clone()) for p in ps]
    oa, ob = Adam(a, lr=1e-3, weight_decay=1e-5), torch.optim.Adam(b, lr=1e-3, weight_decay=1e-5)
    for step in range(5):
        for pa, pb in zip(a, b):
            g = torch.randn_like(pa) * (step + 1)
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(), ob.step()
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6)
    assert int(oa._step.item()) == 5


def test_adam_state_dict_roundtrip_with_torch_adam(dam):
    """optim.Adam speaks torch.optim.Adam's checkpoint format both ways, and a captured step follows param_groups edits."""
    from deep_audio_mixer_amd.optim import Adam
    torch.manual_seed(0)
    ps = [torch.randn(33, 7, device='cuda'), torch.randn(130, device='cuda')]
    a = [torch.nn.Parameter(p.clone()) for p in ps]
    b = [torch.nn.Parameter(p.clone()) for p in ps]
    oa, ob = Adam(a, lr=1e-3, weight_decay=1e-5), torch.optim.Adam(b, lr=1e-3, weight_decay=1e-5)

    def steps(k, scale=1.0):
        for i in range(k):
            for pa, pb in zip(a, b):
                gr = torch.randn_like(pa) * scale
                pa.grad, pb.grad = gr.clone(), gr.clone()
            oa.step(), ob.step()
    steps(3)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa['param_groups'][0]['params'] == sb['param_groups'][0]['params'] == [0, 1]
    for i in (0, 1):
        assert float(sa['state'][i]['step']) == float(sb['state'][i]['step']) == 3.0
        assert torch.allclose(sa['state'][i]['exp_avg'], sb['state'][i]['exp_avg'], rtol=1e-5, atol=1e-8)
        assert torch.allclose(sa['state'][i]['exp_avg_sq'], sb['state'][i]['exp_avg_sq'], rtol=1e-5, atol=1e-10)
    # torch -> ours: a fresh pair continues identically from torch's checkpoint
    a2 = [torch.nn.Parameter(p.detach().clone()) for p in b]
    o2 = Adam(a2, lr=5e-4, weight_decay=0.0)
    o2.load_state_dict(sb)
    assert o2.param_groups[0]['lr'] == 1e-3 and int(o2._step.item()) == 3
    for pa, pb in zip(a2, b):
        gr = torch.randn_like(pa)
        pa.grad, pb.grad = gr.clone(), gr.clone()
    o2.step(), ob.step()
    for pa, pb in zip(a2, b):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
    # ours -> torch
    b3 = [torch.nn.Parameter(p.detach().clone()) for p in a]
    o3 = torch.optim.Adam(b3, lr=1e-3, weight_decay=1e-5)
    o3.load_state_dict(oa.state_dict())
    assert float(o3.state[b3[0]]['step']) == 3.0
    # an LR edit reaches the kernel (hyper-parameters live in a device tensor the launch reads)
    oa.param_groups[0]['lr'] = 0.0
    w = a[0].detach().clone()
    steps(1)
    assert torch.equal(a[0].detach(), w)


def _ddp_graph_worker(rank, world, port, out_dir):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import distributed as ddist
    from deep_audio_mixer_amd.engine import TrainStep
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    torch.cuda.set_device(0)
    ddist.init_process_group('gloo')
    torch.manual_seed(20)
    model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
    ddist.broadcast_module(model)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=world)
    step = TrainStep(model, opt, 2, 16 * 1024, 2, batch=2, use_graph=True)
    assert step.staged and opt.n_buckets == 2
    g = torch.Generator(device='cuda').manual_seed(3)
    stems = 0.1 * torch.randn((4, 2, 16 * 1024, 2), generator=g, device='cuda')
    idx = ddist.shard_indices(4, rank, world)
    step.load_batch(stems[idx], stems[idx].sum(1))
    state = {k: v.clone() for k, v in model.state_dict().items()}
    flat0 = opt._flat.clone()
    step.capture(warmup=1)
    assert len(step._graphs) == 3
    # rewind to the initial replica, then ONE replayed step
    model.load_state_dict(state)
    opt._flat.copy_(flat0), opt._exp_avg.zero_(), opt._exp_avg_sq.zero_(), opt._step.zero_()
    loss = step().item()
    torch.cuda.synchronize()
    torch.save({'params': opt._flat.cpu(), 'grad': opt.flat_grad.cpu(), 'loss': loss}, os.path.join(out_dir, 'd%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_train_step_two_rank_graphs_match_single_process(dam, tmp_path):
    """The N-rank step as bench.py runs it -- graph A1, async all-reduce of the deep bucket beside graph A2, all-reduce
    of the shallow bucket, graph B (Adam) -- against a single process that runs the two micro-batches eagerly, sums the
    gradients and takes the same Adam step.  Two ranks share the box's one GPU (gloo carries the buckets)."""
    import socket
    import torch.multiprocessing as mp
    from deep_audio_mixer_amd import features
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_graph_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'd0.pt'), torch.load(tmp_path / 'd1.pt')
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['grad'], r1['grad'])
    torch.manual_seed(20)
    model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=2)       # 1/2 folded into the update
    g = torch.Generator(device='cuda').manual_seed(3)
    stems = 0.1 * torch.randn((4, 2, 16 * 1024, 2), generator=g, device='cuda')
    state = {k: v.clone() for k, v in model.state_dict().items()}
    total = None
    for idx in ([0, 2], [1, 3]):
        model.load_state_dict(state)
        opt.zero_grad()
        x = features.stft_logmag(stems[idx].reshape(4, 16 * 1024, 2)).view(2, 2, 1025, 17)
        gt = features.stft_logmag(stems[idx].sum(1))
        model.forward_mse(x, gt)[0].backward()
        gsum = opt.gather_grads().clone()
        total = gsum if total is None else total + gsum
    model.load_state_dict(state)
    opt._grad.copy_(total)
    opt.launch_update()
    want_g, want_p = total.cpu(), opt._flat.cpu()
    assert torch.allclose(r0['grad'], want_g, rtol=1e-4, atol=1e-6 * want_g.abs().max())
    # Adam's first update is ~lr*sign(g): compare where the gradient is not rounding-level
    solid = want_g.abs() > 1e-4 * want_g.abs().max()
    assert torch.allclose(r0['params'][solid], want_p[solid], rtol=0, atol=2e-5)


def test_train_step_graph_equals_eager(dam):
    """The hipGraph-captured step replays the same arithmetic as the eager sequence."""
    from deep_audio_mixer_amd.engine import TrainStep
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    losses = []
    for use_graph in (False, True):
        torch.manual_seed(1)
        model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
        opt = Adam(model.parameters(), weight_decay=1e-5)
        step = TrainStep(model, opt, 2, 16 * 1024, 2, batch=2, use_graph=use_graph)
        g = torch.Generator(device='cuda').manual_seed(3)
        stems = 0.1 * torch.randn((2, 2, 16 * 1024, 2), generator=g, device='cuda')
        step.load_batch(stems, stems.sum(1))
        step.capture(warmup=2)
        losses.append([step().item() for _ in range(3)])
    np.testing.assert_allclose(losses[0], losses[1], rtol=1e-4)
    assert losses[0][-1] < losses[0][0]


def test_train_step_bind_clips_reads_resident_batches_in_place(dam):
    """bind_clips(): the captured step follows the front-end's address word to whichever resident batch it points at --
    bitwise the losses of load_clips() (the copy into the static input) over a rotation of batches, back and forth."""
    from deep_audio_mixer_amd.engine import TrainStep
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    g = torch.Generator(device='cuda').manual_seed(5)
    pool = 0.1 * torch.randn((6, 3, 16 * 1024, 2), generator=g, device='cuda')      # three batches of two clips, mix last
    pool[:, 2] = pool[:, :2].sum(1)
    order = [0, 2, 1, 1, 0]
    losses = []
    for bind in (False, True):
        torch.manual_seed(1)
        model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
        step = TrainStep(model, Adam(model.parameters(), weight_decay=1e-5), 2, 16 * 1024, 2, batch=2, use_graph=True)
        step.load_clips(pool[:2])
        step.capture(warmup=2)
        out = []
        for k in order:
            (step.bind_clips if bind else step.load_clips)(pool[2 * k:2 * k + 2])
            out.append(step().item())
        if bind:        # and back to the static buffer
            step.load_clips(pool[4:6])
            out.append(step().item())
        else:
            step.load_clips(pool[4:6])
            out.append(step().item())
        losses.append(out)
    assert losses[0] == losses[1]
    with pytest.raises(ValueError):
        step.bind_clips(pool[:2].cpu())


def test_train_step_bind_rotation_walks_resident_batches(dam):
    """bind_rotation(): the captured front-end indexes a device table of batch addresses with the optimizer's device-side step
    count (DAM_PCM_ROTATE) -- nothing is re-pointed between the replays; bitwise the losses of bind_clips() per step, from any
    starting entry, and bind_clips() afterwards returns to the single-batch form."""
    from deep_audio_mixer_amd.engine import TrainStep
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    g = torch.Generator(device='cuda').manual_seed(6)
    pool = 0.1 * torch.randn((6, 3, 16 * 1024, 2), generator=g, device='cuda')      # three batches of two clips, mix last
    pool[:, 2] = pool[:, :2].sum(1)
    batches = [pool[0:2], pool[2:4], pool[4:6]]
    order = [1, 2, 0, 1, 2, 0, 1] + [2, 2]
    losses = []
    for rotate in (False, True):
        torch.manual_seed(1)
        model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
        step = TrainStep(model, Adam(model.parameters(), weight_decay=1e-5), 2, 16 * 1024, 2, batch=2, use_graph=True)
        step.load_clips(batches[0])
        step.capture(warmup=2)
        out = []
        if rotate:
            step.bind_rotation(batches, first=1)
        for k in order[:7]:
            if not rotate:
                step.bind_clips(batches[k])
            out.append(step().item())
        for k in order[7:]:                       # back to the single-batch form
            step.bind_clips(batches[k])
            out.append(step().item())
        losses.append(out)
        step.close()
    assert losses[0] == losses[1]
    with pytest.raises(ValueError):
        step.bind_rotation([])
    with pytest.raises(ValueError):
        step.bind_rotation([pool[0:2].double()])


def test_step_mark_orders_a_copy_stream_inside_a_captured_step(dam):
    """include/dam_hip.h, dam_step_mark_*: (a) a mark recorded inside a captured graph is re-recorded by every replay and a stream
    outside the graph that waits for it runs BETWEEN the kernels around the mark (never before: 0); (b) TrainStep(copy_mark=True)
    fed by BatchStager(gate=step.copy_mark) -- uploads from page-locked memory timed by the mark, three staging buffers, host-side
    waits -- gives bitwise the losses of the same batches read HBM-resident, and the mark does not change the step."""
    from deep_audio_mixer_amd import staging
    from deep_audio_mixer_amd.engine import TrainStep
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    a = torch.zeros(1 << 26, device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    seen = torch.zeros(6, dtype=torch.int32, device='cuda')
    mark = staging.StepMark()

    def body():
        a.add_(1.0)
        flag.fill_(1)
        mark.record()
        for _ in range(8):
            a.add_(1.0)
        flag.fill_(2)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    side = torch.cuda.Stream()
    for k in range(6):
        flag.zero_()
        g.replay()
        with torch.cuda.stream(side):
            mark.wait(side)
            seen[k:k + 1].copy_(flag)
        torch.cuda.current_stream().wait_stream(side)
    mark.synchronize()
    torch.cuda.synchronize()
    got = seen.tolist()
    # SAFETY is what is asserted: the waiter never runs before the mark of the replay enqueued last (0).  That it runs between the
    # two kernels around the mark (1) rather than behind the graph (2) is timing -- how soon the other queue gets its turn -- and is
    # measured where it matters, on the training step (tools/copy_gate_probe.py, profiles/r05_sync_cost_probe.txt: the upload
    # starts 1.95-2.02 ms into the 4.13 ms step)
    assert all(v in (1, 2) for v in got), got

    g2 = torch.Generator(device='cuda').manual_seed(5)
    pool = 0.1 * torch.randn((8, 3, 16 * 1024, 2), generator=g2, device='cuda')      # four batches of two clips, mix last
    pool[:, 2] = pool[:, :2].sum(1)
    host = torch.empty(pool.shape, dtype=pool.dtype, pin_memory=True)
    host.copy_(pool)
    losses = []
    for kind in ('resident', 'resident+mark', 'streamed', 'streamed+mark'):
        torch.manual_seed(1)
        model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
        step = TrainStep(model, Adam(model.parameters(), weight_decay=1e-5), 2, 16 * 1024, 2, batch=2, use_graph=True,
                         copy_mark='mark' in kind)
        assert (step.copy_mark is not None) == ('mark' in kind)
        step.load_clips(pool[:2])
        step.capture(warmup=2)
        st = staging.BatchStager(host, 2, 'cuda', gate=step.copy_mark) if 'streamed' in kind else None
        out = []
        for k in range(9):
            step.bind_clips(st.next() if st is not None else pool[2 * (k % 4):2 * (k % 4) + 2])
            out.append(step().clone())      # (the loss tensor is the graph's own: one address)
        losses.append([float(v) for v in torch.stack(out).cpu()])
        step.close()
    assert losses[0] == losses[1] == losses[2] == losses[3]


def test_mix_song_smooth_matches_oracle(dam):
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    from deep_audio_mixer_amd.inference_utils import mix_song_smooth
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    sr, n_chunks = 44100, 12
    rng = np.random.default_rng(8)
    tracks = {t: 0.1 * rng.standard_normal((2, sr * n_chunks + 777)) for t in ('bass', 'drums', 'mix')}
    torch.manual_seed(2)
    ref = models_ref.RefResNet18(n_stems=2, input_shape=(1025, 44)).eval()
    model = ResNet18(n_stems=2, input_shape=(1025, 44))
    model.load_state_dict(ref.state_dict())
    model = model.cuda().eval()
    d = MultitrackAudioDataset.from_arrays({'x': {t: v.T for t, v in tracks.items()}}, chunk_length=1, sr=sr,
                                           tracklist=['bass', 'drums', 'mix'])
    mixed, raw, smooth = mix_song_smooth(d, model, tracks, chunk_length=1, sr=sr)

    def model_fn(feats):
        with torch.no_grad():
            return torch.cat(ref(torch.from_numpy(feats))[1], 1)[0].numpy()
    torch.set_num_threads(16)
    mixed_r, raw_r, smooth_r = inference_ref.mix_song_smooth(model_fn, tracks, ['bass', 'drums'], 1, sr)
    for t in ('bass', 'drums'):
        assert len(raw[t]) == n_chunks - 1
        np.testing.assert_allclose(raw[t], raw_r[t], rtol=2e-4)
        np.testing.assert_allclose(smooth[t], smooth_r[t], rtol=2e-4)
        assert mixed[t].shape == tracks[t].shape and mixed[t].dtype == np.float64
        np.testing.assert_allclose(mixed[t], mixed_r[t], rtol=2e-4, atol=1e-9)
    # the callers' next step (sum of the mixed stems, per-channel peak normalise) as one fused pass
    from deep_audio_mixer_amd.inference_utils import mix_song_to_master
    master, raw_m, _ = mix_song_to_master(d, model, tracks, chunk_length=1, sr=sr)
    want = np.sum(np.array([mixed_r[t] for t in ('bass', 'drums')]), axis=0)
    want = want / np.abs(want).max(axis=1, keepdims=True)              # librosa.util.normalize(track_sum, axis=1)
    assert master.shape == want.shape and raw_m['bass'] == raw['bass']
    np.testing.assert_allclose(master, want, rtol=2e-4, atol=2e-6)     # peak is 1: cancelling stem sums need an absolute term
    assert np.allclose(np.abs(master).max(axis=1), 1.0)
    # training-mode models are applied chunk by chunk, as the reference loop does (per-call batch statistics)
    model.train()
    _, raw_t, _ = mix_song_smooth(d, model, tracks, chunk_length=1, sr=sr)
    assert len(raw_t['bass']) == n_chunks - 1 and not np.allclose(raw_t['bass'], raw['bass'])


def _ddp_gpu_worker(rank, world, port, out_dir):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import distributed as ddist
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    torch.cuda.set_device(0)
    ddist.init_process_group('gloo')                      # one GPU on the box: both ranks share it, gloo carries the bucket
    torch.manual_seed(10 + rank)                          # different replicas before the broadcast
    model = ResNet18(n_stems=2, input_shape=(129, 24)).cuda().train()
    ddist.broadcast_module(model)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=world)
    x, gt = model_input(4, 2, 129, 24, seed=77)
    idx = ddist.shard_indices(4, rank, world)
    loss = model.forward_mse(torch.from_numpy(x[idx]).cuda(), torch.from_numpy(gt[idx]).cuda())[0]
    loss.backward()
    opt.step()                                            # gather -> all-reduce(sum) -> Adam with 1/world folded in
    torch.save({'flat_grad': opt.flat_grad.cpu(), 'params': opt._flat.cpu(), 'loss': loss.item()},
               os.path.join(out_dir, 'g%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_step_two_ranks_one_gpu(dam, tmp_path):
    """an N-rank step == a single-process step over N micro-batches with the gradients averaged and local
    BatchNorm statistics.  Two ranks share the box's one GPU (gloo transport; RCCL needs one GPU per rank)."""
    import socket
    import torch.multiprocessing as mp
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'g0.pt'), torch.load(tmp_path / 'g1.pt')
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['flat_grad'], r1['flat_grad'])
    # single process: same initial replica (rank 0's seed), the two micro-batches one after the other
    torch.manual_seed(10)
    model = ResNet18(n_stems=2, input_shape=(129, 24)).cuda().train()
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    x, gt = model_input(4, 2, 129, 24, seed=77)
    total = None
    state = {k: v.clone() for k, v in model.state_dict().items()}
    for idx in ([0, 2], [1, 3]):
        model.load_state_dict(state)                      # BN running stats: each replica starts from the same buffers
        opt.zero_grad()
        model.forward_mse(torch.from_numpy(x[idx]).cuda(), torch.from_numpy(gt[idx]).cuda())[0].backward()
        g = opt.gather_grads().clone()
        total = g if total is None else total + g
    want = total.cpu()                                    # rank buckets hold the SUM; Adam divides by the world size
    assert torch.allclose(r0['flat_grad'], want, rtol=1e-4, atol=1e-6 * want.abs().max())


def test_feature_cache_roundtrip(dam, tmp_path):
    """data/dataset.py:213-268: features pre-computed to {song}_FEATURES/*.npy equal the on-the-fly ones."""
    import wave
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    sr, n = 8000, 8000 * 3 + 123
    rng = np.random.default_rng(2)
    song = tmp_path / 'S' / 'S_STEMS_JOINED'
    song.mkdir(parents=True)
    for name in ('S_STEM_BASS.wav', 'S_STEM_DRUMS.wav', 'S_STEM_VOCALS.wav', 'S_STEM_OTHER.wav', '../S_MIX.wav'):
        x = (rng.uniform(-0.5, 0.5, (n, 2)) * 32767).astype('<i2')
        with wave.open(str(song / name), 'wb') as w:
            w.setnchannels(2), w.setsampwidth(2), w.setframerate(sr)
            w.writeframes(x.tobytes())
    live = MultitrackAudioDataset(str(tmp_path), chunk_length=1, sr=sr)
    assert len(live) == 3
    live._precompute_features()
    assert sorted(os.listdir(tmp_path / 'S' / 'S_FEATURES'))[:2] == ['0_gt_1s.npy', '0_train_1s.npy']
    cached = MultitrackAudioDataset(str(tmp_path), chunk_length=1, sr=sr, compute_features=False)
    for i in range(3):
        a, b = live[i], cached[i]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and b[0].shape == (4, 1025, 8)
    np.random.seed(0)
    aug = MultitrackAudioDataset(str(tmp_path), chunk_length=1, sr=sr, compute_features=False, augment_data=True)[1][0]
    off = (aug - cached[1][0]).flatten(1)
    assert torch.allclose(off, off[:, :1].expand_as(off), atol=1e-4)          # one dB offset per stem (:170-179)
    assert off[:, 0].abs().max() <= 20 * np.log10(1.4) + 1e-3


def test_trainer_captured_step_equals_eager_loop(dam, tmp_path, monkeypatch, capsys):
    """ModelTrainer.fit runs the loop body as one hipGraph replay from the third same-shape batch on: same
    loss list as the all-eager loop, ragged last batch and a second epoch included; loaders without __len__ are accepted."""
    from deep_audio_mixer_amd.model_trainer import ModelTrainer
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    monkeypatch.chdir(tmp_path)
    os.mkdir('weights')
    shape = (2, 1025, 17)
    batches = [tuple(torch.from_numpy(a) for a in model_input(3, *shape, seed=40 + k)) for k in range(6)]
    batches.append(tuple(torch.from_numpy(a) for a in model_input(2, *shape, seed=50)))        # ragged last batch
    val = batches[:1]
    runs = []
    for graph in (False, True):
        torch.manual_seed(4)
        model = ResNet18(n_stems=2, input_shape=shape[1:]).cuda().train()
        opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        tr = ModelTrainer(model, torch.nn.MSELoss(), opt, torch.device('cuda'), model_name='g', graph=graph)
        per_batch = []
        real = tr._train_batch
        tr._train_batch = lambda b, real=real, acc=per_batch: acc.append(real(b).item()) or torch.tensor(acc[-1])
        tl, vl = tr.fit(batches, val, 0, 2)
        assert (tr.graph_steps, tr.eager_steps) == ((10, 4) if graph else (0, 14))     # 2 eager + 4 graph + ragged, then 6 + ragged
        runs.append((list(per_batch), tl, vl, model.state_dict()['bn1.running_mean'].clone()))
        if graph:      # a generator loader (what MultitrackAudioDataset.iter_batches is), epoch mean over the batches seen
            t2, _ = tr.fit((b for b in batches[:3]), val, 2, 1)
            assert len(t2) == 1 and np.isfinite(t2[0])
            tr.close()
            assert not hasattr(next(model.parameters()), '_dam_grad')
    np.testing.assert_allclose(runs[1][0], runs[0][0], rtol=2e-5)
    np.testing.assert_allclose(runs[1][1], runs[0][1], rtol=2e-5)
    np.testing.assert_allclose(runs[1][2], runs[0][2], rtol=2e-5)
    assert torch.allclose(runs[1][3], runs[0][3], rtol=1e-5, atol=1e-6)
    assert runs[0][0][-1] < runs[0][0][0]
    capsys.readouterr()


def _bench_two_ranks(backend):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'DAM_DIST_BACKEND')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0')
    if backend != 'nccl':
        env.update(DAM_DIST_BACKEND=backend)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--no-roofline', '--no-host-stream', '--repeat', '1'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    cfg = out['config']
    assert out['n_gpus'] == 2 and cfg['world_size'] == 2 and cfg['grad_buckets'] == 2
    # the exchange schedule `value` ran with was chosen by timing both on this node, the same on every rank
    assert cfg['reduce_mode'] in ('overlapped', 'inline') and cfg['allreduce_overlap'] is (cfg['reduce_mode'] == 'overlapped')
    assert set(cfg['reduce_mode_calibration_ms']) == {'overlapped', 'inline'}
    assert cfg['reduce_mode'] == min(cfg['reduce_mode_calibration_ms'], key=cfg['reduce_mode_calibration_ms'].get)
    assert cfg['dist_backend'] == backend and cfg['global_batch'] == 16 and cfg['sync_per_step'] is False
    assert np.isfinite(cfg['final_loss']) and out['value'] > 0 and out['steps'] == 3
    pr = out['per_rank']
    assert len(pr['ms_per_step']) == 2 and pr['ms_per_step_min'] <= pr['ms_per_step_max']
    assert all(w is not None and w >= 0 for w in pr['exposed_allreduce_wait_ms'])
    assert out['repeat']['regions'] == 1
    # the two replicas trained on DIFFERENT clips (seed 1234 + rank) and still hold bit-identical parameters: every step's
    # gradient buckets were summed over both ranks
    assert cfg['replicas_in_sync'] is True and len(cfg['replica_checksums']['flat_params_weighted']) == 2
    assert len(cfg['rank_devices']) == 2
    # one record answers "does the overlap pay on this node": the same three graphs timed with the all-reduces beside graph A2
    # and after it, graph A2's own device time in both, per rank
    ab = out['overlap_ab']
    for label in ('overlapped', 'serialized', 'inline'):
        assert ab[label]['ms_per_step'] > 0 and len(ab[label]['graph_A2_ms']) == 2 and all(v > 0 for v in ab[label]['graph_A2_ms'])
        assert len(ab[label]['exposed_allreduce_wait_ms']) == 2
    assert isinstance(ab['overlap_pays'], bool) and ab['graph_A2_slowdown_from_overlap'] > 0
    assert cfg['dist_timeout_s'] > 0 and 'timed region (3 steps)' in cfg['phases_s']
    return out


def test_bench_two_rank_spawn_path_gloo_rehearsal(dam):
    """`python bench.py --gpus 2` as a fresh process: the parent counts GPUs from sysfs, starts two ranks itself
    (torch.distributed.run), the ranks run the staged step (graphs A1 / A2 / B, two async buckets) and rank 0 prints ONE line.
    gloo carries the buckets because this box has one GPU (DAM_DIST_BACKEND=gloo lets two ranks share it); with RCCL the same
    code path runs one rank per GPU.  Not a scaling number -- the line's shape, the staged schedule and the replicas staying
    in sync are what is checked."""
    out = _bench_two_ranks('gloo')
    assert out['config']['rccl_version'] is None


def test_bench_two_rank_spawn_path_rccl(dam):
    """The same spawn path on the DEFAULT backend (RCCL, one rank per GPU) -- runs on the first box that shows two GPUs
    (the count comes from sysfs, bench.visible_gpu_count(): this process starts no second GPU runtime for it).  Checks what
    a SCALE record must show: RCCL saw two ranks on two distinct devices, both buckets travelled, and the replicas --
    fed different clips -- hold bit-identical parameters after the steps."""
    import bench
    n = bench.visible_gpu_count() or 0
    if n < 2:
        pytest.skip('RCCL needs one GPU per rank: %d visible here' % n)
    out = _bench_two_ranks('nccl')
    cfg = out['config']
    assert cfg['rccl_version'] and len({d['device_index'] for d in cfg['rank_devices']}) == 2


def test_trainer_warns_once_when_it_cannot_capture(dam):
    """A criterion or optimizer the captured step does not reproduce falls back to the eager loop -- loudly."""
    from deep_audio_mixer_amd.model_trainer import ModelTrainer
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda()
    with pytest.warns(RuntimeWarning, match='eager'):
        ModelTrainer(model, torch.nn.MSELoss(), torch.optim.SGD(model.parameters(), lr=1e-3), torch.device('cuda'))
    with pytest.warns(RuntimeWarning, match='amsgrad'):
        ModelTrainer(model, torch.nn.MSELoss(), torch.optim.Adam(model.parameters(), amsgrad=True), torch.device('cuda'))
    with pytest.warns(RuntimeWarning, match='criterion'):
        ModelTrainer(model, torch.nn.L1Loss(), torch.optim.Adam(model.parameters()), torch.device('cuda'))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        ModelTrainer(model, torch.nn.L1Loss(), torch.optim.Adam(model.parameters()), torch.device('cuda'), graph=False)
        ModelTrainer(model, torch.nn.MSELoss(), torch.optim.Adam(model.parameters()), torch.device('cuda'))


def test_trainer_pcm_loader_slow_reader_augmentation_torch_adam(dam, tmp_path, monkeypatch, capsys):
    """ModelTrainer.fit over MultitrackAudioDataset.batch_loader(pcm=True) with augment_data=True, a torch.optim.Adam and an
    artificially SLOW reader: the feeder thread's GPU calls (event waits, uploads, the augmentation draw's H2D copy, allocator
    misses) overlap the window in which the trainer captures its step -- captures are thread-local and fenced by
    staging.capture_guard.  The captured, PCM-fed run (front-end inside the graph, gains through the static table) must
    reproduce the all-eager, feature-fed run batch for batch: same augmentation draws (seeded), same losses.  (The eager run
    is handed this package's optim.Adam: both runs then update through the same fused launch -- torch's own foreach Adam
    rounds differently at the 1e-7 level, which lr 1e-3 sign-like first steps amplify to 1e-3 within three batches.)"""
    import time
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    from deep_audio_mixer_amd.model_trainer import ModelTrainer
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    monkeypatch.chdir(tmp_path)
    os.mkdir('weights')
    sr, tracks = 16384, ['a', 'b', 'mix']
    rng = np.random.default_rng(11)
    songs = {}
    for j, chunks in enumerate((7, 6)):
        stems = [(0.1 * rng.standard_normal((chunks * sr, 2))).astype(np.float32) for _ in range(2)]
        songs['s%d' % j] = dict(zip(tracks, stems + [0.7 * stems[0] + 1.2 * stems[1]]))
    runs = []
    for graph in (False, True):
        ds = MultitrackAudioDataset.from_arrays({k: dict(v) for k, v in songs.items()}, chunk_length=1, sr=sr, tracklist=tracks,
                                                seed=5, augment_data=True)
        if graph:
            slow = ds._read_chunk_into

            def slow_read(*a, _real=slow, **k):
                time.sleep(0.01)
                return _real(*a, **k)
            ds._read_chunk_into = slow_read
        train = ds.batch_loader(2, workers=3, pcm=graph)                # 13 items -> 6 full batches + a ragged one
        val = ds.batch_loader(2, indices=[0, 1], workers=2, pcm=graph)
        torch.manual_seed(3)
        model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
        opt = (torch.optim.Adam if graph else Adam)(model.parameters(), lr=1e-3, weight_decay=1e-5)
        tr = ModelTrainer(model, torch.nn.MSELoss(), opt, torch.device('cuda'), model_name='p', graph=graph)
        assert (tr._adopted_from is opt) == graph
        per_batch = []
        real = tr._train_batch
        tr._train_batch = lambda b, real=real, acc=per_batch: acc.append(real(b).item()) or torch.tensor(acc[-1])
        tl, vl = tr.fit(train, val, 0, 2)
        if graph:
            assert (tr.graph_steps, tr.eager_steps) == (10, 4)
            assert tr._step is not None and tr._step.gain is not None and not tr._step.from_features
        runs.append((per_batch, tl, vl))
        tr.close()
    np.testing.assert_allclose(runs[1][0], runs[0][0], rtol=2e-5)
    np.testing.assert_allclose(runs[1][1], runs[0][1], rtol=2e-5)
    # (the validation pass re-reads items 0 and 1: a fresh augmentation draw per read, the same in both runs)
    np.testing.assert_allclose(runs[1][2], runs[0][2], rtol=2e-5)
    capsys.readouterr()


def test_train_step_missing_gradient_raises_before_the_update(dam):
    """A bound parameter that backward never reaches: the first eager step raises BEFORE the all-reduce / Adam launch, so
    parameters and moments are untouched (the poison used to reach Adam), the slots are unbound again, and a
    genuinely non-finite gradient is not mistaken for a missing one."""
    from deep_audio_mixer_amd.engine import TrainStep
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.optim import Adam
    torch.manual_seed(0)
    model = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
    orphan = torch.nn.Parameter(torch.ones(5, device='cuda'))
    opt = Adam(list(model.parameters()) + [orphan], lr=1e-3)
    before = opt._flat.clone()
    step = TrainStep(model, opt, 2, batch=2, feature_shape=(1025, 17), use_graph=False)
    x, gt = (torch.from_numpy(a).cuda() for a in model_input(2, 2, 1025, 17, seed=1))
    step.load_features(x, gt)
    with pytest.raises(RuntimeError, match='wrote no gradient'):
        step.capture(warmup=1)
    assert torch.equal(opt._flat, before) and int(opt._step.item()) == 0
    assert float(opt._exp_avg.abs().max()) == 0.0 and not hasattr(orphan, '_dam_grad')
    # a NaN that backward itself produced is a gradient, not a missing slot
    model2 = ResNet18(n_stems=2, input_shape=(1025, 17)).cuda().train()
    opt2 = Adam(model2.parameters(), lr=1e-3)
    step2 = TrainStep(model2, opt2, 2, batch=2, feature_shape=(1025, 17), use_graph=False)
    bad = x.clone()
    bad[0, 0, 0, 0] = float('nan')
    step2.load_features(bad, gt)
    step2.capture(warmup=1)            # must not raise 'wrote no gradient'
    step2.close()


def _write_wav_song(root, name, n, sr, rng):
    import wave
    song = root / name / (name + '_STEMS_JOINED')
    song.mkdir(parents=True)
    for fn in ('%s_STEM_BASS.wav', '%s_STEM_DRUMS.wav', '%s_STEM_VOCALS.wav', '%s_STEM_OTHER.wav', '../%s_MIX.wav'):
        x = (rng.uniform(-0.5, 0.5, (n, 2)) * 32767).astype('<i2')
        with wave.open(str(song / (fn % name)), 'wb') as w:
            w.setnchannels(2), w.setsampwidth(2), w.setframerate(sr)
            w.writeframes(x.tobytes())


@pytest.mark.parametrize('augment', [False, True])
def test_notebook_loader_cell_with_six_workers_matches_num_workers_0(dam, tmp_path, monkeypatch, capsys, augment):
    """training.ipynb cells 4, 6, 9, 11-13 AS WRITTEN: ``DataLoader(d_train, batch_size, shuffle=False,
    num_workers=6, pin_memory=True, drop_last=False, timeout=0, worker_init_fn=None)`` over MultitrackAudioDataset
    (data/dataset.py:270-292) on WAV files, ``torch.optim.Adam(model.parameters(), weight_decay=1e-5)``,
    ``ModelTrainer(model, criterion, optimizer, device).fit(train_loader, val_loader, 0, 2)``.  The six workers decode on the
    host and touch no GPU API; the batches arrive page-locked, are uploaded beside the running step and feed the PCM-bound
    captured step.  Same loss list -- per batch, per epoch, validation -- as the ``num_workers=0`` loader (whose items are
    CUDA feature tensors; ``pin_memory=True`` is kept there too), augmentation draws included."""
    from torch.utils.data import DataLoader
    from deep_audio_mixer_amd.data.dataset import DeviceBatch, HostPcmBatch, MultitrackAudioDataset
    from deep_audio_mixer_amd.model_trainer import ModelTrainer
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    sr = 16384
    rng = np.random.default_rng(21)
    _write_wav_song(tmp_path, 'A', sr * 8 + 100, sr, rng)
    _write_wav_song(tmp_path, 'B', sr * 7 + 5, sr, rng)
    monkeypatch.chdir(tmp_path)
    os.mkdir('weights')
    device = torch.device('cuda')
    runs = []
    for workers in (0, 6):
        d_train = MultitrackAudioDataset(str(tmp_path), songlist=['A', 'B'], chunk_length=1, sr=sr, seed=321, normalize=False,
                                         compute_features=True, augment_data=augment)
        d_val = MultitrackAudioDataset(str(tmp_path), songlist=['B'], chunk_length=1, sr=sr, seed=321, normalize=False,
                                       compute_features=True, augment_data=False)
        assert len(d_train) == 15
        train_loader = DataLoader(d_train, batch_size=2, shuffle=False, num_workers=workers, pin_memory=True,
                                  drop_last=False, timeout=0, worker_init_fn=None)
        val_loader = DataLoader(d_val, batch_size=2, shuffle=False, num_workers=workers, pin_memory=True,
                                drop_last=False, timeout=0, worker_init_fn=None)
        first = next(iter(train_loader))
        assert type(first) is (HostPcmBatch if workers else DeviceBatch)
        if workers:
            assert first.clips.is_pinned() and first.clips.dtype == torch.int16 and tuple(first.clips.shape) == (2, 5, sr, 2)
            d_train.set_epoch(0)            # (the peek above took the first draw of items 0 and 1)
        else:
            assert first[0].is_cuda and tuple(first[0].shape) == (2, 4, 1025, 17) and len(first) == 2
            d_train.set_epoch(0)
        torch.manual_seed(3)
        model = ResNet18(n_stems=4, input_shape=(1025, 17)).to(device).train()
        criterion = torch.nn.MSELoss()
        optimizer = torch.optim.Adam(model.parameters(), weight_decay=1e-5)
        trainer = ModelTrainer(model, criterion, optimizer, device)
        per_batch = []
        real = trainer._train_batch
        trainer._train_batch = lambda b, real=real, acc=per_batch: acc.append(real(b).item()) or torch.tensor(acc[-1])
        train_loss, val_loss = trainer.fit(train_loader, val_loader, 0, 2)
        assert (trainer.graph_steps, trainer.eager_steps) == (12, 4)          # 8 batches: 2 eager + 5 captured + ragged; 7 + ragged
        assert trainer._step.from_features == (workers == 0)
        assert (trainer._step.gain is not None) == (augment and workers > 0)
        runs.append((per_batch, train_loss, val_loss))
        trainer.close()
    np.testing.assert_allclose(runs[1][0], runs[0][0], rtol=2e-5)
    np.testing.assert_allclose(runs[1][1], runs[0][1], rtol=2e-5)
    np.testing.assert_allclose(runs[1][2], runs[0][2], rtol=2e-5)
    assert len(runs[0][0]) == 16 and all(np.isfinite(runs[0][0]))
    capsys.readouterr()


def test_caller_owned_loop_over_worker_batches(dam):
    """training_ignite.ipynb cell 12 (``train_features, gt_features = batch``; ``model(train_features.float().to(device))``)
    over a DataLoader with workers: unpacking a HostPcmBatch uploads it and runs ONE front-end launch; the pair equals the
    items of the GPU-owning process bit for bit; features.batch_features(batch) is the same as a function and also takes
    the (features, target) pairs of the other loaders."""
    from torch.utils.data import DataLoader
    from deep_audio_mixer_amd import features
    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    rng = np.random.default_rng(4)
    sr = 16000
    songs = {'s%d' % i: {t: (0.1 * rng.standard_normal((int(d * sr), 2))).astype(np.float32)
                         for t in ('bass', 'drums', 'vocals', 'other', 'mix')} for i, d in enumerate([3.2, 2.1])}
    d = MultitrackAudioDataset.from_arrays(songs, chunk_length=1, sr=sr)
    items = [d[i] for i in range(len(d))]
    loader = DataLoader(d, batch_size=2, shuffle=False, num_workers=2, pin_memory=True)
    device = torch.device('cuda')
    seen = 0
    for batch in loader:
        train_features, gt_features = batch
        x, gt = train_features.float().to(device), gt_features.float().to(device)
        assert x.is_cuda and x.dtype == torch.float32
        x2, gt2 = features.batch_features(batch)
        assert x2.data_ptr() != 0 and torch.equal(x2, x) and torch.equal(gt2, gt)
        for j in range(x.shape[0]):
            assert torch.equal(x[j], items[seen][0]) and torch.equal(gt[j], items[seen][1])
            seen += 1
    assert seen == len(d) == 5
    cpu_pair = (items[0][0][None].cpu().double(), items[0][1][None].cpu().double())      # the reference's own item type
    x3, gt3 = features.batch_features(cpu_pair)
    assert x3.is_cuda and x3.dtype == torch.float32 and torch.equal(x3[0], items[0][0]) and torch.equal(gt3[0], items[0][1])


def test_trainer_follows_load_state_dict_on_the_adopted_adam(dam):
    """the usual resume order -- build the trainer, THEN ``optimizer.load_state_dict(checkpoint)`` -- replaces the
    caller's param_groups[0] dict and moment tensors.  The trainer notices before its next batch, loads the moments and
    the step count into the fused buffers and re-shares them: the next update is the one torch.optim.Adam would make
    from the checkpoint, and lr edits on the caller's object still reach the fused launch."""
    from deep_audio_mixer_amd.model_trainer import ModelTrainer
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    shape = (2, 1025, 17)
    batches = [tuple(torch.from_numpy(a).cuda() for a in model_input(2, *shape, seed=60 + k)) for k in range(4)]
    torch.manual_seed(8)
    model = ResNet18(n_stems=2, input_shape=shape[1:]).cuda().train()
    opt = torch.optim.Adam(model.parameters(), weight_decay=1e-5)
    tr = ModelTrainer(model, torch.nn.MSELoss(), opt, torch.device('cuda'))
    for b in batches[:3]:
        tr._train_batch(b)
    tr._push_adopted_state()
    import copy
    ckpt_opt = copy.deepcopy(opt.state_dict())
    ckpt_model = copy.deepcopy(model.state_dict())
    tr._train_batch(batches[3])                                      # the continuation to reproduce
    torch.cuda.synchronize()
    want = tr.optimizer._flat.clone()
    tr.close()
    # resume: fresh model + optimizer + trainer, the checkpoint loaded AFTER the trainer exists
    torch.manual_seed(9)
    model2 = ResNet18(n_stems=2, input_shape=shape[1:]).cuda().train()
    opt2 = torch.optim.Adam(model2.parameters(), weight_decay=1e-5)
    tr2 = ModelTrainer(model2, torch.nn.MSELoss(), opt2, torch.device('cuda'))
    model2.load_state_dict(ckpt_model)
    opt2.load_state_dict(ckpt_opt)
    assert opt2.param_groups[0] is not tr2.optimizer.param_groups[0]          # the hazard: torch swapped the dict
    tr2._train_batch(batches[3])
    torch.cuda.synchronize()
    assert int(tr2.optimizer._step.item()) == 4
    assert opt2.param_groups[0] is tr2.optimizer.param_groups[0]
    assert torch.allclose(tr2.optimizer._flat, want, rtol=1e-5, atol=1e-7)
    p0 = tr2.optimizer._params[0]
    assert opt2.state[p0]['exp_avg'].data_ptr() == tr2.optimizer._exp_avg.data_ptr()
    opt2.param_groups[0]['lr'] = 0.0                                           # still one optimizer
    before = tr2.optimizer._flat.clone()
    tr2._train_batch(batches[0])
    torch.cuda.synchronize()
    assert torch.equal(tr2.optimizer._flat, before)
    tr2.close()
    assert float(opt2.state[p0]['step']) == 5.0
