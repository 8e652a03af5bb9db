"""Oracle for the two training loops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates gridnext/training.py on the CPU with the quirks that change numbers:
  train_spotwise :11-98   zero_grad before every batch (:54); loss.item()*batch_size (:70);
                          epoch loss / len(dataset) (:73); best-val snapshot (:79-84)
  train_gridwise :101-209 patch_classifier.eval() each phase (:126); NO zero_grad before the
                          first backward; mask labels>0, labels-1 (:155-157); loss/accum_iters (:159);
                          step iff batch_ind % accum_iters == 0 (:166) [so batch 0 steps alone];
                          optional f_opt (:169-171); running_loss uses the divided loss (:174);
                          acc over the foreground count (:179); model + .opt checkpoints (:187-195)
Both return (model, val_history, train_history) after restoring the best-val weights.
Pinned by tests/golden/spotwise_*.npz and gridwise_*.npz (reference loops run by tools/gen_golden.py).
"""
import copy
import os

import torch

from .masked_ce import masked_ce


def _to(inputs, device):
    if isinstance(inputs, list):
        return [t.to(device) for t in inputs]
    return inputs.to(device)


def train_spotwise(model, dataloaders, criterion, optimizer, num_epochs=10, outfile=None, log=None):
    device = torch.device("cpu")
    model.to(device)
    hist = {'train': [], 'val': []}
    best_loss, best_wts = float('inf'), copy.deepcopy(model.state_dict())
    for epoch in range(num_epochs):
        for phase in ('train', 'val'):
            model.train(phase == 'train')
            loss_sum, n_right = 0.0, 0
            for inputs, labels in dataloaders[phase]:
                inputs, labels = _to(inputs, device), labels.to(device)
                optimizer.zero_grad()
                with torch.set_grad_enabled(phase == 'train'):
                    out = model(inputs)
                    loss = criterion(out, labels)
                    if phase == 'train':
                        loss.backward()
                        optimizer.step()
                loss_sum += loss.item() * labels.size(0)
                n_right += int((out.argmax(1) == labels).sum())
            n = len(dataloaders[phase].dataset)
            epoch_loss, epoch_acc = loss_sum / n, n_right / n
            if log is not None:
                log.append((epoch, phase, epoch_loss, epoch_acc))
            hist[phase].append(epoch_loss)
            if phase == 'val' and epoch_loss < best_loss:
                best_loss, best_wts = epoch_loss, copy.deepcopy(model.state_dict())
                if outfile is not None:
                    torch.save(model.state_dict(), outfile)
    model.load_state_dict(best_wts)
    return model, hist['val'], hist['train']


def train_gridwise(model, dataloaders, criterion, optimizer, num_epochs=10, outfile=None,
                   f_opt=None, accum_iters=1, log=None):
    device = torch.device("cpu")
    model.to(device)
    hist = {'train': [], 'val': []}
    best_loss, best_wts = float('inf'), copy.deepcopy(model.state_dict())
    for epoch in range(num_epochs):
        for phase in ('train', 'val'):
            model.train(phase == 'train')
            model.patch_classifier.eval()
            loss_sum, n_right, n_fg = 0.0, 0, 0
            for batch_ind, (inputs, labels) in enumerate(dataloaders[phase]):
                inputs, labels = _to(inputs, device), labels.to(device)
                with torch.set_grad_enabled(phase == 'train'):
                    out = model(inputs)
                    assert out.shape[2:] == labels.shape[1:], "Output tensor does not match label dimensions!"
                    C = out.shape[1]
                    flat = out.permute(0, 2, 3, 1).reshape(-1, C)
                    lab = labels.reshape(-1)
                    keep = lab > 0
                    z, t = flat[keep], lab[keep] - 1
                    loss = criterion(z, t) / accum_iters
                    if phase == 'train':
                        loss.backward()
                        if batch_ind % accum_iters == 0:
                            optimizer.step()
                            optimizer.zero_grad()
                            if f_opt is not None:
                                f_opt.step()
                                f_opt.zero_grad()
                loss_sum += loss.item() * labels.size(0)
                n_right += int((z.argmax(1) == t).sum())
                n_fg += int(t.numel())
            epoch_loss = loss_sum / len(dataloaders[phase].dataset)
            epoch_acc = n_right / n_fg
            if log is not None:
                log.append((epoch, phase, epoch_loss, epoch_acc))
            hist[phase].append(epoch_loss)
            if phase == 'val' and epoch_loss < best_loss:
                best_loss, best_wts = epoch_loss, copy.deepcopy(model.state_dict())
                if outfile is not None:
                    torch.save(model.state_dict(), outfile)
                    opt_state = optimizer.state_dict() if f_opt is None else \
                        {'g_opt': optimizer.state_dict(), 'f_opt': f_opt.state_dict()}
                    torch.save(opt_state, os.path.splitext(outfile)[0] + ".opt")
    model.load_state_dict(best_wts)
    return model, hist['val'], hist['train']


def masked_step_loss(out, labels, accum_iters=1):
    """Convenience: the loop's loss on one batch via the standalone masked-CE oracle."""
    return masked_ce(out, labels, accum_iters)[0]
