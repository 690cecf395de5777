"""Host logic of the trainer on CPU (HIP ops replaced by the oracle primitives, tests/fake_backend.py):
BASELINE.json configs[0] plumbing run (16 synthetic utterances, batch 2, 1 epoch, checkpoint + resume),
the collate contract, the hparams singleton and the template config."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from fake_backend import fake_ops  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SMALL_HP = """
architecture = 'text-mel'; model = 'Fastspeech2'; comment = ''
save_dir = {save!r}; train_script = {script!r}; test_script = ''; lengths_file = ''
mean_file = None; var_file = None; spm_model = None
vocab_size = 152; mel_dim = 80; amp = False; optimizer = 'Noam'; warmup_step = 4000; warmup_factor = 1.0
max_seqlen = None; batch_size = 2; max_epoch = 1; save_per_epoch = 1; clip = 1.0
loaded_epoch = None; loaded_dir = None
encoder_type = 'transformer'; decoder_type = 'transformer'
d_model_encoder = 32; n_layer_encoder = 1; n_head_encoder = 2; ff_conv_kernel_size_encoder = 9
d_model_decoder = 32; n_layer_decoder = 1; n_head_decoder = 2; ff_conv_kernel_size_decoder = 1
concat_after_encoder = False; concat_after_decoder = False
postnet_pred = True; reduction_rate = 1; dropout = 0.1
nbins = 256; f0_min = 71.0; f0_max = 799.8; energy_min = 0.0; energy_max = 403.8
pitch_pred = True; energy_pred = True; is_multi_speaker = False; different_spk_emb_samespeaker = False
num_workers = 0; log_every = 4
"""


def test_collate_contract(tmp_path):
    from transformer_tts_amd.datasets import datasets_fastspeech2 as D
    script = D.write_synthetic_corpus(str(tmp_path / "corpus"), n_utt=5)
    hp = SimpleNamespace(mel_dim=80, mean_file=None, var_file=None, spm_model=None, is_multi_speaker=False,
                         tail_alignment="_alignment")
    ds = D.TrainDatasets(script, hp)
    assert len(ds) == 5
    b = D.collate_fn([ds[i] for i in range(3)])
    assert len(b) == 16
    text, mel, pos_text, pos_mel, tl, ml, stop, spk, f0, en, ali = b[:11]
    B, L, T = 3, int(tl.max()), int(ml.max())
    assert text.shape == (B, L) and text.dtype == torch.int64 and mel.shape == (B, T, 80) and mel.dtype == torch.float32
    assert pos_text.dtype == pos_mel.dtype == ali.dtype == torch.int64 and f0.dtype == en.dtype == torch.float32
    for i in range(B):
        n, m = int(tl[i]), int(ml[i])
        assert int(ali[i].sum()) == m                      # sum of durations == mel length
        assert torch.all(mel[i, m:] == -0.5) and torch.all(stop[i, m:] == 1.0) and torch.all(stop[i, :m] == 0.0)
        assert torch.all(text[i, n:] == 0) and torch.all(f0[i, m:] == 0) and torch.all(ali[i, n:] == 0)
        assert pos_text[i, :n].tolist() == list(range(1, n + 1)) and torch.all(pos_mel[i, m:] == 0)
    assert spk is None and b[11] is None and b[12] is None and b[13] is None and b[15] == [None] * B
    s = D.NumBatchSampler(ds, 2, shuffle=False)
    assert list(s) == [[0, 1], [2, 3], [4]]
    w = D.DistributedSamplerWrapper(s, num_replicas=2, rank=1, shuffle=False)
    assert list(w) == [[2, 3], [0, 1]] and len(w) == 2     # padded by repetition to a multiple of the world size
    # default (shuffle=True, seed 0, as the reference's DistributedSampler base): the ranks split one permutation of the
    # batch list, the same one every epoch (tests/test_data_golden.py pins it to the reference's own shards)
    shards = [list(D.DistributedSamplerWrapper(s, num_replicas=2, rank=r)) for r in (0, 1)]
    assert sorted(map(tuple, shards[0] + shards[1]))[1:] == sorted(map(tuple, [[0, 1], [2, 3], [4]])) or \
        sorted(set(map(tuple, shards[0] + shards[1]))) == sorted(map(tuple, [[0, 1], [2, 3], [4]]))


def test_hparams_singleton_and_template(tmp_path):
    from transformer_tts_amd.utils import HParams
    from transformer_tts_amd.utils.utils import fill_variables, get_learning_rate
    h = HParams()
    with pytest.raises(AttributeError, match="not configured"):
        h.foo
    with pytest.raises(FileNotFoundError):
        h.configure(tmp_path / "missing.py")
    with pytest.raises(ValueError):
        (tmp_path / "x.txt").write_text("a=1")
        h.configure(tmp_path / "x.txt")
    h.configure(os.path.join(ROOT, "config", "hparams_template.py"))
    with pytest.raises(RuntimeError, match="reconfigure"):
        h.configure(os.path.join(ROOT, "config", "hparams_template.py"))
    with pytest.raises(AttributeError, match="does not have"):
        h.no_such_key
    fill_variables(h, verbose=False)
    assert h.d_model_encoder == 256 and h.n_layer_decoder == 4 and h.mel_dim == 80 and h.spk_emb_architecture == ""
    assert h.dropout_variance_adaptor == 0.5 and h.fix_mask is None
    from transformer_tts_amd.train_fastspeech2 import build_model
    h.d_model_encoder = h.d_model_decoder = 32      # keep the constructor cheap; all template keys are consumed
    m = build_model(h)
    assert len(m.state_dict()) == 217                # 205 parameters + 4 x 3 BN buffers
    # Noam schedule of the reference (utils/utils.py:204-215)
    assert get_learning_rate(1, 256, 1.0, 4000) == pytest.approx(4000 ** -1.5 * 256 ** -0.5)
    assert get_learning_rate(8000, 256, 1.0, 4000) == pytest.approx(8000 ** -0.5 * 256 ** -0.5)


def test_plumbing_epoch_checkpoint_and_resume(fake_ops, tmp_path, capsys):
    from transformer_tts_amd.datasets import datasets_fastspeech2 as D
    from transformer_tts_amd import train_fastspeech2 as T
    from transformer_tts_amd.utils import HParams
    from transformer_tts_amd.utils.utils import fill_variables
    script = D.write_synthetic_corpus(str(tmp_path / "synthetic16"), n_utt=16)
    hp_file = tmp_path / "hparams.py"
    save = str(tmp_path / "ckpt")
    hp_file.write_text(SMALL_HP.format(save=save, script=script))
    hp = HParams()
    hp.configure(hp_file)
    fill_variables(hp, verbose=False)
    os.makedirs(save, exist_ok=True)
    args = SimpleNamespace(n_gpus=0)
    T.run_training(0, args, hp, None)
    out = capsys.readouterr().out
    assert "EPOCH 1 end" in out and "loss_total" in out and "step 4 / 8" in out
    sd = torch.load(os.path.join(save, "network.epoch1"), weights_only=True)
    assert "encoder.layers.0.ff.f_1.weight" in sd and sd["encoder.layers.0.ff.f_1.weight"].shape == (128, 32, 9)
    osd = torch.load(os.path.join(save, "network.optimizer.epoch1"), weights_only=True)
    assert int(osd["state"][0]["step"]) == 8
    # resume: epoch counter and Adam state come back (reference :428-446)
    hp.loaded_epoch, hp.loaded_dir, hp.max_epoch = 1, save, 2
    T.run_training(0, args, hp, None)
    out = capsys.readouterr().out
    assert "epoch 1 loaded" in out and "EPOCH 2 end" in out
    osd2 = torch.load(os.path.join(save, "network.optimizer.epoch2"), weights_only=True)
    assert int(osd2["state"][0]["step"]) == 16


def test_lengths_batch_sampler_frame_budget(tmp_path):
    """LengthsBatchSampler (reference datasets_fastspeech2.py:749-813): batches in dataset order under the padded-frame
    budget max_len * size <= n_lengths, batch ORDER shuffled, lengths file built when missing and reused afterwards."""
    import random
    from types import SimpleNamespace
    from transformer_tts_amd.datasets import datasets_fastspeech2 as D
    script = D.write_synthetic_corpus(str(tmp_path / "corpus"), n_utt=16, seed=5)
    hp = SimpleNamespace(mel_dim=80, mean_file=None, var_file=None, lengths_file=str(tmp_path / "lengths_built.npy"))
    ds = D.TrainDatasets(script, hp)
    lengths = np.array([ds[i]["mel_length"] for i in range(len(ds))])
    budget = int(lengths.max()) * 3
    random.seed(3)
    s = D.LengthsBatchSampler(ds, budget, hp, lengths_file=None, shuffle=True)
    assert np.array_equal(np.load(hp.lengths_file), lengths), "the lengths file is built from the dataset"
    batches = sorted(list(s), key=lambda b: b[0])
    flat = [i for b in batches for i in b]
    assert flat == list(range(flat[-1] + 1)) and len(flat) >= len(ds) - 1, "consecutive indices, at most the last one dropped"
    for b in batches:
        assert lengths[b].max() * len(b) <= budget
        nxt = b[-1] + 1
        if nxt < len(ds) and b is not batches[-1]:
            assert max(lengths[b].max(), lengths[nxt]) * (len(b) + 1) > budget, "a batch closes only when the next one does not fit"
    s2 = D.LengthsBatchSampler(ds, budget, hp, lengths_file=hp.lengths_file, shuffle=False, reverse=True)
    assert [b[0] for b in s2] == sorted((b[0] for b in batches), reverse=True)
    assert len(s2) == len(batches)
    # the collate function takes such a ragged batch
    out = D.collate_fn([ds[i] for i in batches[0]])
    assert out[1].shape[0] == len(batches[0]) and out[1].shape[1] == lengths[batches[0]].max()


def test_epoch_with_frame_budget_batching(fake_ops, tmp_path, capsys):
    """hp.batch_size = None, hp.max_seqlen set (reference train_fastspeech2.py:340-341): LengthsBatchSampler drives one
    epoch with a different batch shape at (almost) every step."""
    from transformer_tts_amd.datasets import datasets_fastspeech2 as D
    from transformer_tts_amd import train_fastspeech2 as T
    from transformer_tts_amd.utils import HParams
    from transformer_tts_amd.utils.utils import fill_variables
    script = D.write_synthetic_corpus(str(tmp_path / "synthetic16"), n_utt=16)
    hp_file = tmp_path / "hparams.py"
    save = str(tmp_path / "ckpt")
    hp_file.write_text(SMALL_HP.format(save=save, script=script).replace("batch_size = 2", "batch_size = None")
                       + f"\nmax_seqlen = 400\nlog_every = 1\nlengths_file = {str(tmp_path / 'lengths.npy')!r}\n")
    hp = HParams()
    hp.configure(hp_file)
    fill_variables(hp, verbose=False)
    assert hp.batch_size is None and hp.max_seqlen == 400
    os.makedirs(save, exist_ok=True)
    T.run_training(0, SimpleNamespace(n_gpus=0), hp, None)
    out = capsys.readouterr().out
    assert "lengths_file is not exists. Make..." in out and "EPOCH 1 end" in out
    sizes = {int(l.split("=")[1]) for l in out.splitlines() if l.startswith("batch size")}
    assert len(sizes) > 1, f"frame-budget batches should differ in size, got {sizes}"
    assert os.path.exists(str(tmp_path / "lengths.npy"))


def test_module_surface_functions_of_the_reference_trainer(fake_ops):
    """npeak_mask / create_masks (all three task branches) / mse_loss_arelbo / loss_mel"""
    import numpy as np
    import torch.nn.functional as F
    from types import SimpleNamespace
    from transformer_tts_amd import train_fastspeech2 as T
    m = T.npeak_mask(4)
    assert m.shape == (1, 4, 4) and m.dtype == torch.bool
    assert np.array_equal(m[0].cpu().numpy(), np.triu(np.ones((4, 4)), k=1) == 0)
    pos_s, pos_t = torch.tensor([[1, 2, 3, 0]]), torch.tensor([[1, 2, 0]])
    s, t = T.create_masks(pos_s, pos_t, task="fastspeech2")
    assert s.shape == (1, 1, 4) and t.shape == (1, 1, 3) and t.tolist() == [[[True, True, False]]]
    s, t = T.create_masks(pos_s.to(T.DEVICE), pos_t.to(T.DEVICE))
    assert t.shape == (1, 3, 3) and t[0].cpu().tolist() == [[True, False, False], [True, True, False], [True, True, False]]
    assert T.create_masks(pos_s, None)[1] is None
    a, b = torch.randn(3, 7, 5), torch.randn(3, 7, 5)
    assert torch.allclose(T.mse_loss_arelbo(a, b), 0.5 * 35 * torch.log(torch.mean((a - b) ** 2)))
    pred, y = torch.randn(2, 9, 80), torch.randn(2, 9, 80)
    hp = SimpleNamespace(channel_weight=(2.0, 0.5))
    assert abs(T.loss_mel(hp, pred, y).item() - F.l1_loss(pred, y).item()) < 1e-6
    ref = 2.0 * F.l1_loss(pred[:, :, :20], y[:, :, :20]) + 0.5 * F.l1_loss(pred[:, :, 20:], y[:, :, 20:])
    assert abs(T.loss_mel(hp, pred, y, channel_wise=True).item() - ref.item()) < 1e-6


AR_HP = """
architecture = 'text-mel'; model = 'Transformer'; comment = ''
save_dir = {save!r}; train_script = {script!r}; test_script = ''; lengths_file = {lengths!r}
mean_file = None; var_file = None; spm_model = None
vocab_size = 152; mel_dim = 80; amp = False; optimizer = 'Noam'; warmup_step = 4000; warmup_factor = 1.0
max_seqlen = None; batch_size = 3; max_epoch = 1; save_per_epoch = 1; clip = 1.0; accum_grad = 1
loaded_epoch = None; loaded_dir = None
encoder_type = 'transformer'; decoder_type = 'transformer'
d_model_encoder = 32; n_layer_encoder = 1; n_head_encoder = 2; ff_conv_kernel_size_encoder = 3
d_model_decoder = 32; n_layer_decoder = 1; n_head_decoder = 2; ff_conv_kernel_size_decoder = 1
concat_after_encoder = False; concat_after_decoder = False
reduction_rate = {r}; dropout = 0.1; dropout_prenet = 0.5; dropout_postnet = 0.5; positive_weight = 5.0
gst = False; is_multi_speaker = False; spk_emb_dim = None; spk_emb_architecture = ''
num_workers = 0; log_every = 2
"""


@pytest.mark.parametrize("r", [1, 2])
def test_autoregressive_trainer_runs_on_the_reference_data_path(fake_ops, tmp_path, capsys, r):
    """transformer_tts_amd.train.run_training on a corpus read by datasets_transformer (go frame, mel lengths rounded up to the
    reduction rate, batches sorted by length): with reduction rate 2 and utterances of odd length the (T - r) / r decoder steps must
    cover exactly the T - r target frames -- the FastSpeech2 reader's batches made l1 / bce read past their targets."""
    from transformer_tts_amd import train as T
    from transformer_tts_amd.datasets import datasets_fastspeech2 as D
    from transformer_tts_amd.utils import HParams
    from transformer_tts_amd.utils.utils import fill_variables
    script = D.write_synthetic_corpus(str(tmp_path / "synthetic16"), n_utt=7)
    save = str(tmp_path / "ckpt")
    hp_file = tmp_path / "hparams.py"
    hp_file.write_text(AR_HP.format(save=save, script=script, lengths=str(tmp_path / "lengths_ar.npy"), r=r))
    hp = HParams()
    hp.configure(hp_file)
    fill_variables(hp, verbose=False)
    np.random.seed(3)
    step = T.run_training(hp)
    out = capsys.readouterr().out
    assert step == 1 + 3 and "EPOCH 1 end" in out and "loss_total" in out          # 7 utterances in batches of 3: 3 iterations
    assert os.path.exists(os.path.join(save, "network.epoch1"))
    # a batch whose frame count is not a multiple of r is refused with a clear message instead of reading past the target
    if r == 2:
        from transformer_tts_amd.datasets import datasets_transformer as A
        ds = A.TrainDatasets(script, hp)
        batch = A.make_collate_fn(hp)([ds[0], ds[1]])
        odd = tuple(t[:, :-1].contiguous() if torch.is_tensor(t) and t.dim() >= 2 and i in (1, 3, 6) else t for i, t in enumerate(batch))
        model = T.build_model(hp)
        from transformer_tts_amd.optim import FusedAdam
        with pytest.raises(ValueError, match="multiple of hp.reduction_rate"):
            T.train_step(model, FusedAdam(model), 1, odd, hp)


def test_fused_adam_loads_a_reference_optimizer_state_without_entries_for_gradientless_parameters():
    """the reference's torch.optim.Adam holds no state for parameters that never received a gradient (the post-net convolutions of the
    autoregressive model: postnets.py prev_version=False): such a `network.optimizer.epochN` must load -- zero moments for those
    parameters, step and moments of the others restored, and the file FusedAdam writes back loads into torch's Adam"""
    from transformer_tts_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Conv1d(4, 4, 3), torch.nn.Linear(16, 2))
    ref = torch.optim.Adam(net.parameters(), lr=1e-3)
    for _ in range(3):
        ref.zero_grad()
        net[2](net[0](torch.randn(5, 8))).sum().backward()       # the Conv1d gets no gradient: no state entry
        ref.step()
    sd = ref.state_dict()
    assert set(sd["state"].keys()) == {0, 1, 4, 5}
    opt = FusedAdam(net)
    opt.load_state_dict(sd)
    assert opt.t == 3
    for i, (p, o) in enumerate(zip(opt.arena.params, opt.arena.offsets)):
        m = opt.m[o:o + p.numel()].view_as(p)
        if i in sd["state"]:
            assert torch.equal(m, sd["state"][i]["exp_avg"])
        else:
            assert float(m.abs().max()) == 0.0
    back = torch.optim.Adam(net.parameters(), lr=1e-3)
    back.load_state_dict(opt.state_dict())
    assert int(back.state_dict()["state"][0]["step"]) == 3


def test_bench_launcher_glue_runs_two_ranks_without_a_gpu():
    """`python bench.py --gpus 2` from a plain shell: bench.launch_ranks builds the torch.distributed.run command line, the two rank
    processes rendezvous over 127.0.0.1, run the barrier / MAX-over-ranks protocol of the timed region and rank 0 prints ONE JSON
    line.  FS2_BENCH_STUB=1 replaces the GPU work by a sleep (gloo): the launcher glue is what is under test."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, FS2_BENCH_STUB="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["frames_all_ranks"] == 2 * 3000.0 and line["local_rank"] == 0
    assert line["ms_per_step"] >= 0.9 * 20.0, "the MAX over ranks must report the slower rank's time"
    # the driver's own form: started under torch.distributed.run, bench.py must not launch a second set of ranks
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1
