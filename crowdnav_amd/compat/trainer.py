"""Replay memory and SGD trainer on the reference's interfaces (crowd_nav/utils/memory.py:4-28,
trainer.py:8-71).  They stay plain PyTorch (ROCm) exactly as BASELINE configs[4] asks: the rollouts come from the
HIP engine, the optimizer from torch.  When the reference is importable its own classes can be used instead."""
import logging

import torch.nn as nn
import torch.optim as optim
from torch.utils.data import DataLoader, Dataset


class ReplayMemory(Dataset):
    """Ring buffer of (state [H, D] float32, value [1] float32) pairs."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.memory = []
        self.position = 0

    def push(self, item):
        if self.position < len(self.memory):
            self.memory[self.position] = item
        else:
            self.memory.append(item)
        self.position = (self.position + 1) % self.capacity

    def is_full(self):
        return len(self.memory) == self.capacity

    def __getitem__(self, index):
        return self.memory[index]

    def __len__(self):
        return len(self.memory)

    def clear(self):
        self.memory = []


class Trainer(object):
    """SGD(momentum 0.9) + MSE on the value network; optimize_epoch for imitation learning, optimize_batch for RL."""

    def __init__(self, model, memory, device, batch_size):
        self.model = model
        self.device = device
        self.criterion = nn.MSELoss().to(device)
        self.memory = memory
        self.data_loader = None
        self.batch_size = batch_size
        self.optimizer = None

    def set_learning_rate(self, learning_rate):
        logging.info('Current learning rate: %f', learning_rate)
        self.optimizer = optim.SGD(self.model.parameters(), lr=learning_rate, momentum=0.9)

    def _loader(self):
        if self.optimizer is None:
            raise ValueError('Learning rate is not set!')
        if self.data_loader is None:
            self.data_loader = DataLoader(self.memory, self.batch_size, shuffle=True)
        return self.data_loader

    def _fit(self, inputs, values):
        self.optimizer.zero_grad()
        loss = self.criterion(self.model(inputs.to(self.device)), values.to(self.device))
        loss.backward()
        self.optimizer.step()
        return loss.data.item()

    def optimize_epoch(self, num_epochs):
        loader = self._loader()
        average_epoch_loss = 0
        for epoch in range(num_epochs):
            epoch_loss = sum(self._fit(inputs, values) for inputs, values in loader)
            average_epoch_loss = epoch_loss / len(self.memory)
            logging.debug('Average loss in epoch %d: %.2E', epoch, average_epoch_loss)
        return average_epoch_loss

    def optimize_batch(self, num_batches):
        loader = self._loader()
        losses = 0
        for _ in range(num_batches):
            inputs, values = next(iter(loader))  # a fresh shuffled iterator per batch, as trainer.py:57 does
            losses += self._fit(inputs, values)
        average_loss = losses / num_batches
        logging.debug('Average loss : %.2E', average_loss)
        return average_loss
