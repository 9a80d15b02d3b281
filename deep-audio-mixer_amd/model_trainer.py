"""ModelTrainer -- drop-in for the reference's model_trainer.py:5-67 (same constructor, same ``fit``
signature and return value, same stdout lines and checkpoint file names), driving the HIP models.

Differences that do not change results: when ``criterion`` is ``nn.MSELoss()`` (what every notebook passes)
the loss and its gradient come from the model's fused ``forward_mse`` (one pass over the features instead of
masked-sum + MSE + two backward passes); any other criterion goes through ``criterion(masked, gt)`` exactly as
in the reference.  Like the reference, ``fit`` never calls ``model.train()``/``model.eval()``.
With ``torch.distributed`` initialised only rank 0 prints and writes checkpoints.
"""
import os

import torch


class ModelTrainer:
    def __init__(self, model, criterion, optimizer, device, model_name='scalar2d'):
        self.weights_dir = './weights'
        self.model_name = model_name
        self.model = model
        self.optimizer = optimizer
        self.criterion = criterion
        self.device = device
        self._fused = (type(criterion) is torch.nn.MSELoss and criterion.reduction == 'mean'
                       and hasattr(model, 'forward_mse'))

    @staticmethod
    def _is_main():
        return not (torch.distributed.is_available() and torch.distributed.is_initialized()) or \
            torch.distributed.get_rank() == 0

    def _loss(self, train_features, gt_features):
        x, gt = train_features.to(self.device), gt_features.to(self.device)
        if self._fused:
            return self.model.forward_mse(x, gt)[0]
        masked, _ = self.model(x)
        return self.criterion(masked, gt)

    def _validate_epoch(self, val_loader):
        running_val_loss = 0.0
        with torch.no_grad():
            for i, batch in enumerate(val_loader):
                train_features, gt_features = batch
                running_val_loss += self._loss(train_features, gt_features).item()
        return running_val_loss / len(val_loader)

    def _train_epoch(self, train_loader):
        running_loss = 0.0
        for i, batch in enumerate(train_loader):
            self.optimizer.zero_grad()
            train_features, gt_features = batch
            loss = self._loss(train_features, gt_features)
            loss.backward()
            self.optimizer.step()
            value = loss.item()
            each_n_batches = 10
            if i % each_n_batches == each_n_batches - 1 and self._is_main():
                print('[%d/%4d] loss: %.3f' % (i + 1, len(train_loader), value))
            running_loss += value
        return running_loss / len(train_loader)

    def fit(self, train_loader, val_loader, start_epoch, num_epochs):
        train_loss = []
        val_loss = []
        main = self._is_main()
        for epoch in range(start_epoch, start_epoch + num_epochs):
            if main:
                print('Epoch {}/{}'.format(epoch, num_epochs - 1))
            avg_epoch_loss = self._train_epoch(train_loader)
            train_loss.append(avg_epoch_loss)
            if main:
                print('Epoch {} train loss: {:.4f}'.format(epoch, avg_epoch_loss))
            avg_epoch_val_loss = self._validate_epoch(val_loader)
            val_loss.append(avg_epoch_val_loss)
            if main:
                print('Epoch {} val loss: {:.4f}'.format(epoch, avg_epoch_val_loss))
                print('-' * 50)
                weights_file = os.path.join(self.weights_dir,
                                            'mixmodel_{}_1s_{:04d}_{:.4f}.pt'.format(self.model_name, epoch, avg_epoch_loss))
                torch.save(self.model.state_dict(), weights_file)
        return train_loss, val_loss
