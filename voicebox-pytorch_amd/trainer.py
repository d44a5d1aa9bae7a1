"""VoiceBoxTrainer-compatible driver over the native data-parallel step.

Mirrors voicebox_pytorch/trainer.py:60-321 for training on LATENTS (the audio codec / raw-wave front end is out of the hot
path's scope): same constructor keywords, `train_step` / `train` / `save` / `load`, the same learning-rate rule (linear warm-up,
then one CosineAnnealingLR step per training step, :231-253), gradient accumulation with a deferred exchange (= accelerator.no_sync
on all but the last micro-batch, :258-272), global-norm clipping (:274-275), Adam(betas=(0.9, 0.99)) (optimizer.py:10-35), rank-0
validation and checkpoints in the reference's format `{'model', 'optim', 'scheduler'}` (:191-197) -- `optim` is a
torch.optim.Adam state_dict over `cfm_wrapper.parameters()` order, so checkpoints interchange with the reference trainer.

No Accelerate: one process per GPU with torch.distributed already initialised by the launcher (or a single process).
"""
import re
from pathlib import Path
from shutil import rmtree

import torch
import torch.distributed as dist
from torch import nn
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import DataLoader, Dataset, random_split

from .dp import TrainStep, WarmupCosineLR
from .model import ConditionalFlowMatcherWrapper


def exists(v):
    return v is not None


def cycle(dl, sampler=None):
    epoch = 0
    while True:
        if sampler is not None:  # DistributedSampler reshuffles only when told the epoch (accelerate's prepared loader does this)
            sampler.set_epoch(epoch)
        for data in dl:
            yield data
        epoch += 1


def checkpoint_num_steps(checkpoint_path):  # trainer.py:44-57
    results = re.findall(r'\d+', str(checkpoint_path))
    return int(results[-1]) if len(results) else 0


def _collate(pad_to_longest):  # data.py:69-101 (tensors only)
    def one(data):
        if pad_to_longest:
            return pad_sequence(list(data), batch_first=True)
        n = min(d.shape[0] for d in data)
        return torch.stack([d[:n] for d in data])

    def inner(batch):
        if not isinstance(batch[0], tuple):
            return (one(batch),)
        return tuple(one(col) for col in zip(*batch))

    return inner


def get_dataloader(ds, pad_to_longest=True, **kwargs):  # data.py:103-105
    return DataLoader(ds, collate_fn=_collate(pad_to_longest), **kwargs)


class VoiceBoxTrainer(nn.Module):
    def __init__(self, cfm_wrapper: ConditionalFlowMatcherWrapper, *, batch_size, dataset: Dataset, num_train_steps=None,
                 num_warmup_steps=None, num_epochs=None, lr=3e-4, initial_lr=1e-5, grad_accum_every=1, wd=0., max_grad_norm=0.5,
                 valid_frac=0.05, random_split_seed=42, log_every=10, save_results_every=100, save_model_every=1000,
                 results_folder='./results', force_clear_prev_results=None, split_batches=False, drop_last=False,
                 accelerate_kwargs: dict = dict(), length_bucket=64):
        # length_bucket (not a reference keyword): batches of a varying-length dataset are padded with masked frames to the next
        # multiple of it, so that training runs on a handful of activation arenas (dp.TrainStep); 0 = exact lengths
        super().__init__()
        assert isinstance(cfm_wrapper, ConditionalFlowMatcherWrapper)
        self.wd = float(wd)  # > 0: AdamW with decay on the ndim >= 2 parameters (get_optimizer, optimizer.py:10-35)
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.distributed else 0
        self.world = dist.get_world_size() if self.distributed else 1
        # split_batches (trainer.py:83,93 -> Accelerator(split_batches=...)): batch_size is the GLOBAL batch of a step, every rank
        # loads batch_size / world of it (False, Accelerate's default: every rank loads its own batch_size samples)
        self.split_batches = bool(split_batches)
        self.rank_batch_size = self.rank_batch(batch_size, self.world, self.split_batches)
        self.cfm_wrapper = cfm_wrapper
        self.register_buffer('steps', torch.Tensor([0]))
        self.batch_size, self.grad_accum_every = batch_size, grad_accum_every
        self.lr, self.initial_lr, self.max_grad_norm = lr, initial_lr, max_grad_norm

        self.ds = dataset
        if valid_frac > 0:  # trainer.py:121-131
            train_size = int((1 - valid_frac) * len(self.ds))
            valid_size = len(self.ds) - train_size
            self.ds, self.valid_ds = random_split(self.ds, [train_size, valid_size],
                                                  generator=torch.Generator().manual_seed(random_split_seed))
            self.print(f'training with dataset of {len(self.ds)} samples and validating with randomly splitted {len(self.valid_ds)} samples')
        else:
            self.valid_ds = self.ds
            self.print(f'training with shared training and valid dataset of {len(self.ds)} samples')
        assert len(self.ds) >= batch_size, 'dataset must have sufficient samples for training'
        assert len(self.valid_ds) >= batch_size, \
            f'validation dataset must have sufficient number of samples (currently {len(self.valid_ds)}) for training'
        assert exists(num_train_steps) or exists(num_epochs), 'either num_train_steps or num_epochs must be specified'
        self.num_train_steps = len(dataset) // batch_size * num_epochs if exists(num_epochs) else num_train_steps
        self.num_warmup_steps = num_warmup_steps if exists(num_warmup_steps) else 0
        self.schedule = WarmupCosineLR(lr, self.num_train_steps, self.num_warmup_steps, initial_lr)
        self.train_step_fn = TrainStep(cfm_wrapper, lr=lr, betas=(0.9, 0.99), eps=1e-8, max_grad_norm=max_grad_norm, wd=self.wd,
                                       lr_schedule=None, length_bucket=length_bucket)

        sampler = None
        if self.world > 1:  # what accelerator.prepare(dl) does: each rank sees its own shard
            sampler = torch.utils.data.distributed.DistributedSampler(self.ds, num_replicas=self.world, rank=self.rank, shuffle=True)
        self.dl = get_dataloader(self.ds, batch_size=self.rank_batch_size, shuffle=sampler is None, sampler=sampler, drop_last=drop_last)
        self.valid_dl = get_dataloader(self.valid_ds, batch_size=batch_size, shuffle=True, drop_last=drop_last)
        self.dl_iter, self.valid_dl_iter = cycle(self.dl, sampler), cycle(self.valid_dl)
        self.log_every, self.save_model_every, self.save_results_every = log_every, save_model_every, save_results_every
        self.results_folder = Path(results_folder)
        if self.is_main and force_clear_prev_results is True and self.results_folder.exists():
            rmtree(str(self.results_folder))
        self.results_folder.mkdir(parents=True, exist_ok=True)

    @staticmethod
    def rank_batch(batch_size, world, split_batches):
        """Samples one rank loads per step: batch_size (Accelerate's default) or batch_size / world (split_batches=True)."""
        if not split_batches:
            return batch_size
        if batch_size % world:
            raise ValueError(f"split_batches=True: batch_size ({batch_size}) must be a round multiple of the number of processes "
                             f"({world})")  # Accelerate's own condition
        return batch_size // world

    # ---- checkpoint format of the reference (trainer.py:191-207)
    def _optim_param_order(self):
        """Parameters in the order the reference's optimizer holds them: cfm_wrapper.parameters(), or -- with weight decay -- the
        ndim >= 2 parameters followed by the others (the two param groups get_optimizer builds, optimizer.py:4-9,24-30)."""
        params = list(self.cfm_wrapper.parameters())
        if self.wd > 0:
            wd_p = [p for p in params if p.ndim >= 2]
            return wd_p + [p for p in params if p.ndim < 2], len(wd_p)
        return params, len(params)

    def _optim_state_dict(self, gathered=False):
        ts = self.train_step_fn
        if not gathered:
            # grad_mode="shard": the moments of a chunk live on its owner; gathering them is a collective, so a direct save() /
            # _optim_state_dict() call must be made by EVERY rank (train_step() gathers on all ranks itself and passes gathered=True)
            ts.gather_optimizer_state()
        fp = ts.fp
        by_param = {id(fp.slots[s]): s for s in fp.order}
        state = {}
        params, n_wd = self._optim_param_order()
        for i, p in enumerate(params):
            s = by_param.get(id(p))
            if s is None or ts.steps == 0:
                continue  # parameters that never received a gradient (null_cond) have no Adam state, as in torch
            o, n = fp.offsets[s], p.numel()
            state[i] = {'step': torch.tensor(float(ts.steps)), 'exp_avg': ts.m[o:o + n].view(p.shape).clone(),
                        'exp_avg_sq': ts.v[o:o + n].view(p.shape).clone()}
        base = dict(lr=self.schedule.cur, betas=tuple(ts.betas), eps=ts.eps, amsgrad=False, maximize=False, foreach=None,
                    capturable=False, differentiable=False, fused=None)
        if self.wd > 0:
            groups = [dict(base, weight_decay=self.wd, params=list(range(n_wd))),
                      dict(base, weight_decay=0, params=list(range(n_wd, len(params))))]
        else:
            groups = [dict(base, weight_decay=0, params=list(range(len(params))))]
        return {'state': state, 'param_groups': groups}

    def _load_optim_state_dict(self, sd):
        ts = self.train_step_fn
        fp = ts.fp
        by_param = {id(fp.slots[s]): s for s in fp.order}
        steps = 0
        params, n_wd = self._optim_param_order()
        # Validate BEFORE copying anything: a checkpoint saved under the other weight-decay grouping (a reference wd = 0
        # checkpoint into a wd > 0 trainer, or the reverse) numbers its parameters differently; torch.optim.load_state_dict raises a
        # ValueError here, and so do we -- a partial, silently mis-assigned Adam state is never left behind.
        want = [n_wd, len(params) - n_wd] if self.wd > 0 else [len(params)]
        got = [len(g['params']) for g in sd['param_groups']]
        if got != want:
            raise ValueError(f"loaded state dict has parameter groups of sizes {got}, this trainer (wd = {self.wd}) has {want}")
        for i, p in enumerate(params):
            st = sd['state'].get(i)
            if st is not None and (tuple(st['exp_avg'].shape) != tuple(p.shape) or tuple(st['exp_avg_sq'].shape) != tuple(p.shape)):
                raise ValueError(f"optimizer state {i} has shape {tuple(st['exp_avg'].shape)}, parameter has {tuple(p.shape)}")
        for i, p in enumerate(params):
            st = sd['state'].get(i)
            s = by_param.get(id(p))
            if st is None or s is None:
                continue
            o, n = fp.offsets[s], p.numel()
            ts.m[o:o + n].copy_(st['exp_avg'].reshape(-1))
            ts.v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps = max(steps, int(float(st['step'])))
        ts.steps = steps

    def _scheduler_state_dict(self):  # loadable by torch's CosineAnnealingLR.load_state_dict
        sc = self.schedule
        return dict(T_max=sc.T, eta_min=0, base_lrs=[sc.lr], last_epoch=sc.sched_epoch, _step_count=sc.sched_epoch + 1,
                    _get_lr_called_within_step=False, _last_lr=[sc.cur], verbose=False)

    def save(self, path, gathered=False):
        """Reference checkpoint format.  With TrainStep(grad_mode="shard") and world size > 1 call it on EVERY rank (only rank 0 needs
        to pass a real path to keep) unless the optimizer state was gathered collectively beforehand (gathered=True)."""
        pkg = dict(model=self.cfm_wrapper.state_dict(), optim=self._optim_state_dict(gathered), scheduler=self._scheduler_state_dict())
        torch.save(pkg, path)

    def load(self, path):
        pkg = self.cfm_wrapper.load(path)
        self.train_step_fn.fp = self.cfm_wrapper.voicebox.flat_params()  # load_state_dict copied into the flat views
        self.train_step_fn._dirty()
        self._load_optim_state_dict(pkg['optim'])
        sc = pkg['scheduler']
        self.schedule.sched_epoch, self.schedule.cur = int(sc['last_epoch']), float(sc['_last_lr'][0])
        # + 1 to start from the next step and avoid overwriting the last checkpoint (trainer.py:206-207)
        self.steps = torch.tensor([checkpoint_num_steps(path) + 1], device=self.steps.device)

    def print(self, msg):
        if self.rank == 0:
            print(msg, flush=True)

    @property
    def device(self):
        return self.cfm_wrapper.device

    @property
    def is_distributed(self):
        return self.world > 1

    @property
    def is_main(self):
        return self.rank == 0

    def warmup(self, step):  # trainer.py:231-235
        if step < self.num_warmup_steps:
            return self.initial_lr + (self.lr - self.initial_lr) * step / self.num_warmup_steps
        return self.lr

    def _model_kwargs(self, batch):
        """(latents,) or (latents, cond_token_ids): the second column feeds a text-conditioned model as semantic ids."""
        x = batch[0]
        if len(batch) > 1 and self.cfm_wrapper.condition_on_text:
            return x, dict(cond_token_ids=batch[1])
        return x, {}

    def train_step(self):  # trainer.py:237-313
        steps = int(self.steps.item())
        lr = self.schedule.rate_for_step(steps)
        ts = self.train_step_fn
        logs = {}
        if self.grad_accum_every == 1:
            x, kw = self._model_kwargs(next(self.dl_iter))
            loss = ts.step(x, lr=lr, **kw)
            logs['loss'] = float(loss)
        else:
            total = 0.
            for i in range(self.grad_accum_every):
                x, kw = self._model_kwargs(next(self.dl_iter))
                if i + 1 < self.grad_accum_every:   # accelerator.no_sync: no exchange (trainer.py:258-272)
                    loss = ts.accumulate(x, 1.0 / self.grad_accum_every, **kw)
                else:                              # last micro-batch: exchange overlapped with ITS backward, then clip + Adam
                    loss = ts.accumulate_last_and_apply(x, 1.0 / self.grad_accum_every, lr=lr, **kw)
                total += float(loss) / self.grad_accum_every
            logs['loss'] = total
        if not steps % self.log_every:
            self.print(f"{steps}: loss: {logs['loss']:0.3f}")
        if self.distributed:
            dist.barrier()
        if self.is_main and not (steps % self.save_results_every):  # rank-0 validation (:289-301)
            x, kw = self._model_kwargs(next(self.valid_dl_iter))
            with torch.inference_mode():
                self.cfm_wrapper.eval()
                ids = kw.get('cond_token_ids')
                valid_loss = self.cfm_wrapper(x.to(self.device), **({'semantic_token_ids': ids.to(self.device)} if ids is not None else {}))
            self.print(f'{steps}: valid loss {float(valid_loss):0.3f}')
            logs['valid_loss'] = float(valid_loss)
        if not (steps % self.save_model_every):
            # grad_mode="shard": the Adam moments of a chunk live on its owner -- collecting them is a COLLECTIVE, so every rank takes
            # part here, before rank 0 alone writes the file (a rank-0-only gather would hang or pair with the other ranks'
            # next-step collectives).  A no-op in all-reduce mode.
            self.train_step_fn.gather_optimizer_state()
        if self.is_main and not (steps % self.save_model_every):
            self.save(str(self.results_folder / f'voicebox.{steps}.pt'), gathered=True)
            self.print(f'{steps}: saving model to {str(self.results_folder)}')
        self.steps += 1
        return logs

    def train(self, log_fn=lambda logs: None):
        while self.steps < self.num_train_steps:
            log_fn(self.train_step())
        self.print('training complete')
