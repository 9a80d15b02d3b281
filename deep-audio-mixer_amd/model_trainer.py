"""ModelTrainer for the HIP models -- same public surface and observable behaviour as the reference's
model_trainer.py:5-67: ``ModelTrainer(model, criterion, optimizer, device, model_name='scalar2d')`` and
``fit(train_loader, val_loader, start_epoch, num_epochs) -> (train_loss, val_loss)``, the same progress lines on stdout
and the same ``./weights/mixmodel_{name}_1s_{epoch:04d}_{loss:.4f}.pt`` checkpoints (the directory must
exist, "_1s_" is literal, the header prints ``num_epochs - 1``).

What is different underneath:
  * with ``criterion = nn.MSELoss()`` (what every notebook passes) loss and gradient come from the model's fused
    ``forward_mse`` (one pass over the features); any other criterion is applied to ``masked`` as in the reference;
  * like the reference, the trainer never switches the model between train and eval mode: validation runs
    under ``torch.no_grad()`` with whatever mode the caller left the model in;
  * under ``torch.distributed`` only rank 0 prints and saves.
"""
import os

import torch

_LOG_EVERY = 10                                   # model_trainer.py:39
_CKPT_PATTERN = 'mixmodel_{}_1s_{:04d}_{:.4f}.pt'  # model_trainer.py:64


def _rank0():
    dist = torch.distributed
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class ModelTrainer:
    def __init__(self, model, criterion, optimizer, device, model_name='scalar2d'):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.device = device
        self.model_name = model_name
        self.weights_dir = './weights'
        fusable = type(criterion) is torch.nn.MSELoss and criterion.reduction == 'mean'
        self._fused = fusable and hasattr(model, 'forward_mse')

    # ---- one batch -> loss tensor (on the device)
    def _batch_loss(self, batch):
        feats, target = (t.to(self.device) for t in batch)
        if self._fused:
            return self.model.forward_mse(feats, target)[0]
        return self.criterion(self.model(feats)[0], target)

    def _run(self, loader, train):
        """Mean loss over the loader; one optimisation step per batch when ``train``."""
        total, quiet = 0.0, not _rank0()
        for step, batch in enumerate(loader, start=1):
            if train:
                self.optimizer.zero_grad()
                loss = self._batch_loss(batch)
                loss.backward()
                self.optimizer.step()
            else:
                loss = self._batch_loss(batch)
            value = loss.item()
            if train and step % _LOG_EVERY == 0 and not quiet:
                print('[%d/%4d] loss: %.3f' % (step, len(loader), value))
            total += value
        return total / len(loader)

    def _train_epoch(self, train_loader):
        return self._run(train_loader, True)

    def _validate_epoch(self, val_loader):
        with torch.no_grad():
            return self._run(val_loader, False)

    def fit(self, train_loader, val_loader, start_epoch, num_epochs):
        history = {'train': [], 'val': []}
        talk = _rank0()
        for epoch in range(start_epoch, start_epoch + num_epochs):
            if talk:
                print('Epoch {}/{}'.format(epoch, num_epochs - 1))
            history['train'].append(self._train_epoch(train_loader))
            if talk:
                print('Epoch {} train loss: {:.4f}'.format(epoch, history['train'][-1]))
            history['val'].append(self._validate_epoch(val_loader))
            if talk:
                print('Epoch {} val loss: {:.4f}'.format(epoch, history['val'][-1]))
                print('-' * 50)
                name = _CKPT_PATTERN.format(self.model_name, epoch, history['train'][-1])
                torch.save(self.model.state_dict(), os.path.join(self.weights_dir, name))
        return history['train'], history['val']
