"""Oracle restatement of one reference training step (non-amp branch).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Follows train_fastspeech2.py:116-315 of the
reference: Noam lr, masks, forward, five L1 losses over ALL positions (padding included),
backward, clip_grad_norm_(1.0), Adam(betas=(0.9,0.98), eps=1e-9).
"""
import torch
import torch.nn.functional as F


def create_masks(pos_text, pos_mel):
    """train_fastspeech2.py:55-82, task 'fastspeech2': (pos != 0).unsqueeze(-2)."""
    return (pos_text != 0).unsqueeze(-2), (pos_mel != 0).unsqueeze(-2)


def noam_lr(step, d_model, warmup_factor, warmup_step):
    """utils/utils.py:204-215."""
    return warmup_factor * min(step ** -0.5, step * warmup_step ** -1.5) * (d_model ** -0.5)


def losses(outputs, mel, alignment, f0, energy):
    """train_fastspeech2.py:212-259: nn.L1Loss() (mean over every element) for mel before/after,
    log-duration vs log(alignment+1), pitch and energy; total = their sum."""
    mel_before, mel_after, log_d, p_pred, e_pred = outputs[:5]
    parts = {
        "mel": F.l1_loss(mel_before, mel),
        "post_mel": F.l1_loss(mel_after, mel),
        "duration": F.l1_loss(log_d, torch.log(alignment.to(log_d.dtype) + 1)),
    }
    total = parts["mel"] + parts["post_mel"]
    if p_pred is not None:                  # hp.pitch_pred (:249-252)
        parts["f0"] = F.l1_loss(p_pred, f0)
        total = total + parts["f0"]
    if e_pred is not None:                  # hp.energy_pred (:254-257)
        parts["energy"] = F.l1_loss(e_pred, energy)
        total = total + parts["energy"]
    total = total + parts["duration"]
    return total, parts


def forward_backward(model, batch):
    """Forward + losses + backward on one 16-tuple batch; returns (total, parts, outputs)."""
    text, mel, pos_text, pos_mel, _, _, _, _, f0, energy, alignment = batch[:11]
    src_mask, mel_mask = create_masks(pos_text, pos_mel)
    dt = next(model.parameters()).dtype
    out = model(text, src_mask, mel_mask, alignment, f0.to(dt), energy.to(dt))
    total, parts = losses(out, mel.to(dt), alignment, f0.to(dt), energy.to(dt))
    for p in model.parameters():
        p.grad = None
    total.backward()
    return total, parts, out


def make_optimizer(model):
    """train_fastspeech2.py:411-416."""
    return torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)


def train_step(model, optimizer, step, batch, d_model, warmup_factor=1.0, warmup_step=4000, clip=1.0):
    """One iteration of train_loop (train_fastspeech2.py:116-120,153-154,300-315); returns
    (loss, step + 1)."""
    lr = noam_lr(step, d_model, warmup_factor, warmup_step)
    for g in optimizer.param_groups:
        g["lr"] = lr
    total, parts, _ = forward_backward(model, batch)
    assert not torch.isnan(total), "loss is nan"
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    optimizer.step()
    return total.detach(), step + 1
