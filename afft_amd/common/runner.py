"""MI355X mirror of the reference's ``common/runner.py``: the loss (future CE + past CE + past-feature MSE)
and the per-batch step wrapper.

  MultiDimCrossEntropy  <- common/runner.py:13-37   fused softmax-CE kernel (hard labels with ignore_index=-1,
                                                     or soft/one-hot targets with a boolean row filter)
  BasicLossAccuracy     <- common/runner.py:40-168
  Runner                <- common/runner.py:178-270

Differences that do not change any reduced value: with soft targets the reference drops ignored rows and
returns a shorter vector; here the vector keeps its length, ignored rows are 0 and the kept rows are scaled
by rows/kept, so ``torch.mean`` (Runner._reduce_loss) gives the same number without a device->host sync.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import functional as F_

CLS_MAP_PREFIX = 'cls_map_'
PAST_LOGITS_PREFIX = 'past_'


def accuracy(output, target, topk=(1,)):
    """top-k accuracy over flattened leading dims (common/utils.py:59-86).  The reference returns zeros early when every target is
    negative (`if torch.all(target < 0)`: a device-to-host sync in the middle of every training step); the same zeros fall out of
    the arithmetic -- a predicted class index is never negative, so nothing matches -- and the host never waits."""
    with torch.no_grad():
        output = output.flatten(0, -2)
        target = target.flatten()
        maxk = max(topk)
        batch_size = target.size(0)
        _, pred = output.topk(maxk, 1, True, True)
        pred = pred.t()
        correct = pred.eq(target[None])
        return [correct[:k].flatten().sum(dtype=torch.float32) * (100.0 / batch_size) for k in topk]


class MultiDimCrossEntropy(nn.Module):
    """Flattens the leading dimensions, then per-row cross entropy (reduction='none', ignore_index=-1)."""

    def __init__(self, ignore_index: int = -1, reduction: str = 'none'):
        super().__init__()
        assert ignore_index == -1 and reduction == 'none', "the AFFT path uses ignore_index=-1, reduction='none'"

    def forward(self, inp, tgt, one_hot: bool = False, ignore_index: Union[torch.Tensor, None] = None):
        C = inp.size(-1)
        landing = getattr(inp, "_afft_landing", None)
        # a (clips, frames, C) view with contiguous classes (one half of the merged classifier output) is walked where it lies;
        # anything else is flattened to rows (a copy when the leading dimensions do not collapse)
        inp2 = inp if (inp.dim() == 3 and inp.stride(2) == 1 and inp.dtype == torch.float32) else inp.reshape(-1, C)
        rows = inp.numel() // C
        if not one_hot:
            labels = tgt.reshape(-1).to(torch.int64).contiguous()
            return F_.SoftmaxCE.apply(inp2, labels, None, None, landing)
        soft = tgt.reshape(-1, C).to(torch.float32).contiguous()
        keep = None
        if ignore_index is not None:
            keep = (~ignore_index.reshape(-1)).to(torch.uint8).contiguous()
        row_loss = F_.SoftmaxCE.apply(inp2, None, soft, keep, landing)
        if keep is not None:
            row_loss = row_loss * (float(rows) / keep.sum().clamp(min=1).to(torch.float32))
        return row_loss


class BasicLossAccuracy(nn.Module):
    """acc1 / acc5 / mt5r inputs and the three loss terms."""

    def __init__(self, compute_metrics: bool = True, lazy_host: bool = False):
        super().__init__()
        self.cls_criterion = MultiDimCrossEntropy(ignore_index=-1, reduction='none')
        self.compute_metrics = compute_metrics
        self.lazy_host = lazy_host        # host copies of the logits / labels as LazyHostArray instead of a blocking .cpu()

    @staticmethod
    def reg_criterion(a, b, skip: int = 0):
        """MSELoss(a[:, skip:], b[:, skip:]) (common/runner.py:164-166 slices [:, 1:] first): frames that are contiguous C-vectors
        are walked in place (F_.MSE with a frame range); anything else takes the copying path."""
        B, T, C = a.shape
        if (a.dtype == b.dtype == torch.float32 and a.stride(2) == 1 and b.stride(2) == 1 and a.stride(1) == C and b.stride(1) == C
                and T - skip > 0):
            return F_.MSE.apply(a, b, skip, skip, T - skip)
        a, b = a[:, skip:], b[:, skip:]
        return F_.MSE.apply(a.reshape(-1, C).contiguous(), b.reshape(-1, C).contiguous())

    def forward_future_action(self, logits, tgt_val, mixup_enable, losses, metrics, acc1_key, acc5_key, mt5r_key,
                              loss_key, key_suffix=''):
        losses[loss_key + key_suffix] = self.cls_criterion(logits, tgt_val, one_hot=mixup_enable)
        if not self.compute_metrics:
            return
        sequence_index = 0
        if mixup_enable:
            _vals, inds = torch.topk(tgt_val, 2, dim=1, largest=True, sorted=True)
            rows = torch.arange(tgt_val.shape[0], device=logits.device)
            seq = torch.full_like(rows, sequence_index)
            preds = logits.detach().clone()
            preds[rows, seq, inds[:, 0]] += preds[rows, seq, inds[:, 1]]
            # a device-resident zero: a Python scalar on the right-hand side is staged through pageable host memory, and that
            # copy waits for everything queued in front of it (the whole forward pass)
            preds.index_put_((rows, seq, inds[:, 1]), torch.zeros((), dtype=preds.dtype, device=preds.device))
            labels = inds[:, 0]
        else:
            preds = logits.detach()
            labels = tgt_val.clone()
        if len(labels.shape) == 1:
            labels = labels.unsqueeze(dim=-1)
        if self.lazy_host and preds.is_cuda:
            metrics[mt5r_key + key_suffix] = {'logits': LazyHostArray(preds[:, sequence_index, :].contiguous()),
                                              'labels': LazyHostArray(labels[:, sequence_index].contiguous())}
        else:
            metrics[mt5r_key + key_suffix] = {'logits': preds[:, sequence_index, :].cpu().numpy(),
                                              'labels': labels[:, sequence_index].cpu().numpy()}
        acc1, acc5 = accuracy(preds, labels, topk=(1, min(5, preds.size(-1))))
        metrics[acc1_key + key_suffix] = acc1
        metrics[acc5_key + key_suffix] = acc5

    def forward_past_action(self, past_logits, past_target, mixup_enable, losses, loss_key,
                            past_target_ignore_index=None, key_suffix=''):
        if mixup_enable:
            assert past_logits.shape == past_target.shape
            assert past_target_ignore_index is not None
            loss = self.cls_criterion(past_logits, past_target, one_hot=True, ignore_index=past_target_ignore_index)
        else:
            past_target = past_target.squeeze(-1)
            assert past_logits.shape[:-1] == past_target.shape
            loss = self.cls_criterion(past_logits, past_target)
        losses[loss_key + key_suffix] = loss

    def forward(self, outputs, target, target_subclips, mixup_enable: bool = False,
                target_subclips_ignore_index: Union[Dict, None] = None):
        losses, metrics = {}, {}
        for tgt_type, tgt_val in target.items():
            for modk in outputs[f'logits/{tgt_type}']:
                logits = outputs[f'logits/{tgt_type}'][modk]
                assert len(logits.shape) == 3
                self.forward_future_action(logits, tgt_val, mixup_enable, losses, metrics,
                                           f'acc1_{tgt_type}_{modk}', f'acc5_{tgt_type}_{modk}',
                                           f'mt5r_{tgt_type}_{modk}', f'cls_{tgt_type}_{modk}')
            past_key = f'{PAST_LOGITS_PREFIX}logits/{tgt_type}'
            if past_key in outputs and target_subclips is not None:
                for modk in outputs[past_key]:
                    ign = None if target_subclips_ignore_index is None else target_subclips_ignore_index[tgt_type]
                    self.forward_past_action(outputs[past_key][modk], target_subclips[tgt_type], mixup_enable,
                                             losses, f'past_cls_{tgt_type}_{modk}', ign)
            if 'orig_past' in outputs and 'past_futures' in outputs:
                for modk, upd in outputs['past_futures'].items():
                    if modk not in outputs['orig_past']:
                        continue
                    losses[f'past_reg_{modk}'] = self.reg_criterion(upd, outputs['orig_past'][modk], skip=1)
        return losses, metrics


def get_loss_wts(loss_wts: Dict, key: str) -> float:
    for k, v in loss_wts.items():
        if key.startswith(k):
            return v
    raise ValueError(f'{key} not contained in predefined loss_wts: {loss_wts}')


class _Pending:
    """one non-blocking device-to-host copy into pinned memory + the event that says it has landed"""

    def __init__(self, dev_tensor: torch.Tensor):
        t = dev_tensor.detach()
        if t.is_cuda:
            self.host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self.host.copy_(t, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = t.clone(), None

    def wait(self) -> torch.Tensor:
        if self.event is not None:
            self.event.synchronize()
            self.event = None
        return self.host


class LazyHostArray:
    """A numpy array that is still on its way from the GPU: what `tensor.cpu().numpy()` returns in the reference
    (common/runner.py:82-85, the (B, 3806) logits of the mean-top-5-recall meter), without the host waiting for the GPU inside
    the forward call.  The first numpy use (np.asarray / np.argsort / an attribute / an operator: metric_tracking.py:23-29)
    waits for the copy's event only."""

    def __init__(self, dev_tensor: torch.Tensor):
        self._p = _Pending(dev_tensor)
        self._a = None

    def _arr(self):
        if self._a is None:
            self._a = self._p.wait().numpy()
        return self._a

    def __array__(self, dtype=None, copy=None):
        a = self._arr()
        return a if dtype is None else a.astype(dtype)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._arr(), name)

    def __getitem__(self, i):
        return self._arr()[i]

    def __len__(self):
        return len(self._arr())

    def __eq__(self, o):
        return self._arr() == o

    def __ne__(self, o):
        return self._arr() != o

    __hash__ = None


class LazyScalar:
    """A loss value that is still on its way from the GPU: what `tensor.item()` returns in the reference
    (common/runner.py:211-212), usable wherever the training loop uses that float (AverageMeter.update: val * n, += ;
    formatting; comparisons) -- the first such use waits for the copy and raises the reference's 'The loss is NaN!' then."""

    def __init__(self, pending: _Pending, index: int):
        self._p, self._i, self._v = pending, index, None

    def _val(self) -> float:
        if self._v is None:
            self._v = float(self._p.wait()[self._i])
            if self._v != self._v:
                raise ValueError('The loss is NaN!')
        return self._v

    def __float__(self):
        return self._val()

    def item(self):
        return self._val()

    def __repr__(self):
        return repr(self._val())

    def __format__(self, spec):
        return format(self._val(), spec)

    def __mul__(self, o):
        return self._val() * o

    __rmul__ = __mul__

    def __add__(self, o):
        return self._val() + o

    __radd__ = __add__

    def __sub__(self, o):
        return self._val() - o

    def __rsub__(self, o):
        return o - self._val()

    def __truediv__(self, o):
        return self._val() / o

    def __rtruediv__(self, o):
        return o / self._val()

    def __neg__(self):
        return -self._val()

    def __abs__(self):
        return abs(self._val())

    def __pow__(self, o):
        return self._val() ** o

    def __round__(self, n=None):
        return round(self._val(), n)

    def __bool__(self):
        return bool(self._val())

    def __lt__(self, o):
        return self._val() < float(o)

    def __le__(self, o):
        return self._val() <= float(o)

    def __gt__(self, o):
        return self._val() > float(o)

    def __ge__(self, o):
        return self._val() >= float(o)

    def __eq__(self, o):
        return self._val() == float(o)

    __hash__ = None


class PendingScalars:
    """Loss values on their way to the host: ONE pinned, non-blocking device-to-host copy of all scalars of a step and an
    event, instead of one blocking .item() per loss term (common/runner.py:211-212 syncs the GPU every step).
    result() waits for that copy only (and raises the reference's 'The loss is NaN!' then)."""

    def __init__(self, named: Dict[str, torch.Tensor]):
        self.keys = list(named)
        dev = torch.stack([named[k].detach().float().reshape(()) for k in self.keys])
        if dev.is_cuda:
            self.host = torch.empty(len(self.keys), dtype=torch.float32, pin_memory=True)
            self.host.copy_(dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = dev.clone(), None

    def ready(self) -> bool:
        return self.event is None or self.event.query()

    def result(self) -> Dict[str, float]:
        if self.event is not None:
            self.event.synchronize()
        vals = dict(zip(self.keys, self.host.tolist()))
        if any(v != v for v in vals.values()):
            raise ValueError('The loss is NaN!')
        return vals


class Runner:
    """wrapper class of BasicLossAccuracy, runs on each batch, returns all metrics (common/runner.py:178-270; the reference
    constructs it as Runner(model, device, loss_wts=...), train.py:371).
    async_metrics: what happens to the values the reference fetches with blocking copies inside this call (the .item() of every
    loss term, :211-212; the .cpu().numpy() of the logits for the recall meter, :82-85), i.e. BEFORE backward is enqueued:
      "lazy" (default; AFFT_RUNNER_SYNC=1 turns it off): the SAME keys hold LazyScalar / LazyHostArray values -- non-blocking
          pinned copies that wait for their event at their first use (metric_tracker.update at the end of the iteration,
          train.py:278), so the unchanged loop enqueues forward, backward and the update without the host ever waiting;
      False: the reference's blocking fetches, value for value;
      True: metrics['losses'] is ONE PendingScalars for all terms; the recall meter's arrays are lazy as above."""

    def __init__(self, model, device, loss_wts, compute_metrics: bool = True, async_metrics=None):
        import os
        if async_metrics is None:
            async_metrics = False if os.environ.get("AFFT_RUNNER_SYNC", "0") == "1" else "lazy"
        self.model = model
        self.device = device
        self.loss_acc_fn = BasicLossAccuracy(compute_metrics, lazy_host=bool(async_metrics))
        self.loss_wts = loss_wts
        self.async_metrics = async_metrics

    @staticmethod
    def _reduce_loss(losses, loss_wts, sync: bool = True):
        keys = list(losses)
        wts = [max(float(get_loss_wts(loss_wts, k)), 0.0) for k in keys]      # a weight <= 0 drops the term from the total (runner.py:205-207)
        if not any(w > 0 for w in wts):
            raise RuntimeError("Runner._reduce_loss: every loss weight is <= 0 -- there is nothing to sum (the reference's "
                               "torch.sum(torch.stack([])) fails here too, runner.py:205-207)")
        if 1 <= len(keys) <= 8:
            # dropped terms go in DETACHED: they keep their slot (their mean is still logged) but leave the graph, as in the
            # reference -- no backward work through their branch, and the kernel skips them in the total (0 * NaN would poison it)
            vals = [losses[k] if w > 0 else losses[k].detach() for k, w in zip(keys, wts)]
            loss, means = F_.ReduceLosses.apply(tuple(wts), *vals)      # one launch each way
            losses = {k: means[i] for i, k in enumerate(keys)}
        else:
            losses = {key: torch.mean(val) for key, val in losses.items()}
            loss = torch.sum(torch.stack([w * losses[k] for k, w in zip(keys, wts) if w > 0]))
        if not sync:
            return loss, {k: v.detach() for k, v in losses.items()}
        if torch.isnan(loss):
            raise ValueError('The loss is NaN!')
        losses_metric = {k: v.item() for k, v in losses.items()}
        losses_metric['total_loss'] = loss.item()
        return loss, losses_metric

    def __call__(self, data, mixup_fn: Optional[Callable] = None, mixup_backbone: Optional[bool] = True):
        data, timings = data
        feature_dict = {mod: t.to(self.device, non_blocking=True) for mod, t in data["data_dict"].items()}
        target = {k: v.to(self.device, non_blocking=True) for k, v in data['target'].items()}
        target_subclips = None
        if 'target_subclips' in data:
            target_subclips = {k: v.to(self.device, non_blocking=True) for k, v in data['target_subclips'].items()}
        kwargs = dict(mixup_fn=None, target=target, target_subclips=target_subclips,
                      target_subclips_ignore_index=None)
        if mixup_fn is not None:
            if not mixup_backbone:
                feature_dict, target, target_subclips, ign = mixup_fn(feature_dict, target, target_subclips)
                kwargs.update(target=target, target_subclips=target_subclips, target_subclips_ignore_index=ign)
            else:
                kwargs['mixup_fn'] = mixup_fn
        outputs, out_t = self.model(feature_dict, **kwargs)
        losses, metrics = self.loss_acc_fn(outputs, out_t['target'], out_t['target_subclips'],
                                           mixup_enable=(mixup_fn is not None),
                                           target_subclips_ignore_index=out_t['target_subclips_ignore_index'])
        if self.async_metrics == "lazy":
            loss, dev_losses = self._reduce_loss(losses, self.loss_wts, sync=False)
            dev_losses['total_loss'] = loss.detach()
            keys = list(dev_losses)
            pend = _Pending(torch.stack([dev_losses[k].detach().float().reshape(()) for k in keys]))
            metrics.update({k: LazyScalar(pend, i) for i, k in enumerate(keys)})
        elif self.async_metrics:
            loss, dev_losses = self._reduce_loss(losses, self.loss_wts, sync=False)
            dev_losses['total_loss'] = loss.detach()
            metrics['losses'] = PendingScalars(dev_losses)
        else:
            loss, losses_metric = self._reduce_loss(losses, self.loss_wts)
            metrics.update(losses_metric)
        metrics.update(timings)
        return loss, metrics
