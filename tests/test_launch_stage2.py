"""CPU (build container only), world_size 2, gloo: tools/launch_stage2.py drives the reference's REAL `train_stage2.Trainer`
(/root/reference/train_stage2.py:27-97, imported unmodified) data-parallel.  Only the leaves the container cannot provide are stubbed,
at the model boundary, through the launcher's --hook: the network (a two-layer stand-in with the attributes Trainer touches, including
parameters that never receive a gradient, like the reference's gru16 / gru32), the THuman2.0 data set, and pts2render (there is no GPU
here).  Everything in between is the reference's own loop: DataLoader construction, fetch_data, loss = flow + 0.8 L1 + 0.2 (1 - SSIM)
with the reference's lib/loss.py, GradScaler, clip_grad_norm_, AdamW + OneCycleLR, Logger, save_ckpt.

Checked: every rank ends with bit-identical parameters that differ from the initial ones (the exchange step ran), the ranks were fed
different samples (DistributedSampler), only rank 0 wrote checkpoints / logs, and the module-global `cfg` the Trainer reads exists.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout only exists in the build container")

HOOK = textwrap.dedent('''
    import json, os
    import torch, torch.nn as nn, torch.distributed as dist
    torch.Tensor.cuda = lambda self, *a, **k: self          # no GPU in the build container
    nn.Module.cuda = lambda self, *a, **k: self
    SEEN = []

    class _Raft(nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = nn.BatchNorm2d(3)
            self.unused = nn.Linear(4, 4)                    # constructed, never run: receives no gradient (core/update.py:105-106)
        def freeze_bn(self):
            self.bn.eval()

    class TinyModel(nn.Module):
        def __init__(self, cfg, with_gs_render=True):
            super().__init__()
            self.raft_stereo = _Raft()
            self.head = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 3, 3, padding=1))
        def forward(self, data, is_train=True):
            prev = len(SEEN) - 1                             # the iteration that just ended (train_stage2.py:92: eval when total_steps % eval_freq == 0)
            if EVAL_FREQ and prev > 0 and prev % EVAL_FREQ == 0:
                # rank 0 validated after that iteration; EVERY rank must have waited for it (the launcher's barrier) before starting this one
                assert os.path.exists(EVAL_MARK % prev), "rank %d ran ahead of rank 0's validation pass" % rank
            SEEN.append(int(data["sample_id"][0]))
            x = torch.cat([data["lmain"]["img"], data["rmain"]["img"]], 0)
            y = self.head(self.raft_stereo.bn(x))
            data["_pred"] = y
            flow_loss = (y ** 2).mean()
            return data, flow_loss, {"epe": float(flow_loss.detach())}

    class ToySet(torch.utils.data.Dataset):
        val_boost = 1
        def __init__(self, opt, phase="train"):
            self.n = 16 if phase == "train" else 2
        def __len__(self):
            return self.n
        def __getitem__(self, i):
            g = torch.Generator().manual_seed(100 + i)
            im = lambda: torch.rand(3, 16, 16, generator=g)
            return {"sample_id": i, "lmain": {"img": im()}, "rmain": {"img": im()}, "novel_view": {"img": im()}}

    def fake_pts2render(data, bg_color):
        B = data["novel_view"]["img"].shape[0]
        data["novel_view"]["img_pred"] = torch.sigmoid(data["_pred"][:B] + data["_pred"][B:])
        return data

    TS.RtStereoHumanModel, TS.StereoHumanDataset, TS.pts2render = TinyModel, ToySet, fake_pts2render
    EVAL_FREQ = int(os.environ.get("LAUNCH_TEST_EVAL_FREQ", "0"))
    EVAL_MARK = os.environ["LAUNCH_TEST_OUT"] + ".eval_%d"
    EVAL_RANKS = []

    def toy_eval(self):                                      # stands in for Trainer.run_eval (train_stage2.py:103-139): slow, and rank 0 only
        import time
        EVAL_RANKS.append(rank)
        time.sleep(0.7)
        open(EVAL_MARK % self.total_steps, "w").write("done")

    TS.Trainer.run_eval = toy_eval
    real_train = TS.Trainer.train

    def train_and_verify(self):
        init = [p.detach().clone() for p in self.model.parameters()]
        real_train(self)
        flat = torch.cat([p.detach().reshape(-1) for p in self.model.parameters()])
        moved = float((flat - torch.cat([p.reshape(-1) for p in init])).abs().max())
        both = [None] * world
        dist.all_gather_object(both, (flat.tolist(), SEEN, EVAL_RANKS))
        out = os.environ["LAUNCH_TEST_OUT"]
        if rank == 0:
            json.dump({"identical": both[0][0] == both[1][0], "moved": moved, "seen": [b[1] for b in both], "cfg_global": TS.cfg is cfg,
                       "steps": int(self.total_steps), "eval_ranks": [b[2] for b in both]}, open(out, "w"))

    TS.Trainer.train = train_and_verify
''')


@pytest.mark.parametrize("form", ["checkout", "checkout-eval-inside-the-run", "checkout-stock-ddp", "checkout-stock-ddp-eval-inside-the-run"])
def test_launcher_trains_the_reference_trainer_data_parallel(tmp_path, form):
    ref = REF
    eval_freq = 2 if form.endswith("eval-inside-the-run") else 1000
    hook = tmp_path / "hook.py"
    hook.write_text(HOOK)
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", LAUNCH_TEST_OUT=str(out), LAUNCH_TEST_EVAL_FREQ=str(eval_freq if eval_freq < 1000 else 0))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "tools", "launch_stage2.py"), "--reference", ref, "--backend", "gloo", "--steps", "6", "--exp-root", str(tmp_path / "experiments"),
           "--hook", str(hook)] + (["--ddp"] if "stock-ddp" in form else []) + [   # --ddp: torch's DistributedDataParallel(find_unused_parameters=True)
           "stage1_ckpt", "None", "batch_size", "2", "record.loss_freq", "2", "record.eval_freq", str(eval_freq)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert res["identical"] and res["moved"] > 0 and res["cfg_global"] and res["steps"] == 6
    # validation inside the run (eval_freq 2: after iterations 2 and 4): rank 0 ran it, rank 1 never did -- and waited for it (the forward's assertion)
    assert res["eval_ranks"] == ([[0, 0], []] if eval_freq == 2 else [[], []]), res["eval_ranks"]
    s0, s1 = res["seen"]
    assert len(s0) == 6 and len(s1) == 6 and not (set(s0) & set(s1) and s0 == s1)      # different shards
    line = [l for l in r.stdout.splitlines() if l.startswith("{") and "launch_stage2" in l]
    assert len(line) == 1 and json.loads(line[0])["world_size"] == 2 and json.loads(line[0])["ddp"] is ("stock-ddp" in form)
    exp = list((tmp_path / "experiments").iterdir())
    assert len(exp) == 1
    ckpts = sorted(p.name for p in (exp[0] / "ckpt").iterdir())
    assert any(n.endswith("_final.pth") for n in ckpts) and any(n.endswith("_latest.pth") for n in ckpts)   # rank 0 only wrote them (one set)
    assert (exp[0] / "file" / "cfg.json").exists() and (exp[0] / "logs").exists()
