"""Replay memory and SGD trainer on the reference's interfaces (crowd_nav/utils/memory.py:4-28,
trainer.py:8-71).  They stay plain PyTorch (ROCm) exactly as BASELINE configs[4] asks: the rollouts come from the
HIP engine, the optimizer from torch.  When the reference is importable its own classes can be used instead."""
import logging
import os

import torch
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

    def __getitem__(self, item):
        return self.memory[item]

    def __len__(self):
        return len(self.memory)

    def clear(self):
        self.memory = []


class DeviceReplayMemory(Dataset):
    """The same ring (memory.py:4-28: push order, overwrite from position 0 once full) held as two device tensors —
    states [capacity, H, D] and values [capacity, 1] float32 — so that a batched rollout pushes all its (state, value)
    pairs with one index_copy (push_batch) and the trainer draws batches with one gather (no DataLoader, no per-item
    collate).  Still a Dataset of (state, value) pairs for code that iterates it the reference's way."""

    def __init__(self, capacity, device='cuda:0'):
        self.capacity = int(capacity)
        self.device = torch.device(device)
        self.states = None
        self.values = torch.zeros(self.capacity, 1, dtype=torch.float32, device=self.device)
        self.size = 0
        self.position = 0

    def push_batch(self, states, values):
        """states [n, ...] and values [n] (or [n, 1]) in push order; equivalent to n push() calls."""
        n = int(states.shape[0])
        if n == 0:
            return
        states = states.to(device=self.device, dtype=torch.float32)
        values = values.to(device=self.device, dtype=torch.float32).reshape(n, 1)
        if self.states is None:
            self.states = torch.zeros((self.capacity,) + tuple(states.shape[1:]), dtype=torch.float32,
                                      device=self.device)
        if n > self.capacity:  # only the last `capacity` pushes survive; the ring position still advances by n
            self.position = (self.position + n - self.capacity) % self.capacity
            states, values, n_eff = states[-self.capacity:], values[-self.capacity:], self.capacity
        else:
            n_eff = n
        head = min(n_eff, self.capacity - self.position)  # rows up to the end of the ring: two plain copies, not five kernels
        self.states[self.position:self.position + head].copy_(states[:head])
        self.values[self.position:self.position + head].copy_(values[:head])
        if head < n_eff:  # the rest wraps round
            self.states[:n_eff - head].copy_(states[head:])
            self.values[:n_eff - head].copy_(values[head:])
        self.position = (self.position + n_eff) % self.capacity
        self.size = min(self.capacity, self.size + n)

    def push(self, item):
        state, value = item
        self.push_batch(state.unsqueeze(0), value.reshape(1))

    def is_full(self):
        return self.size == self.capacity

    def __getitem__(self, index):
        if not -self.size <= index < self.size:
            raise IndexError(index)
        return self.states[index], self.values[index]

    def __len__(self):
        return self.size

    def clear(self):
        self.size = 0
        self.position = 0

    def batches(self, batch_size, limit=None):
        """One shuffled pass (DataLoader(shuffle=True) semantics: a random permutation cut into batches, the last one
        partial); limit = number of batches to yield."""
        perm = torch.randperm(self.size, device=self.device)
        stops = range(0, self.size, batch_size)
        for n, start in enumerate(stops):
            if limit is not None and n >= limit:
                return
            idx = perm[start:start + batch_size]
            yield self.states.index_select(0, idx), self.values.index_select(0, idx)


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
        self._graph = None          # hipGraph of one full-batch SGD step (device memory on a GPU only)
        self._graph_failed = os.environ.get('CROWDNAV_AMD_SGD_GRAPH', '1') == '0'

    def set_learning_rate(self, learning_rate):
        logging.info('Current learning rate: %f', learning_rate)
        self.optimizer = optim.SGD(self.model.parameters(), lr=learning_rate, momentum=0.9)
        self._graph = None  # the captured step belongs to the previous optimizer

    def _capture(self, inputs, values):
        """Capture forward + backward + SGD update of one full batch into a graph (the step is ~60 tiny kernels:
        launch-bound).  Warm-up iterations run on a side stream as torch requires; parameters and momentum buffers are
        put back afterwards, so capturing does not train."""
        params = [p for g in self.optimizer.param_groups for p in g['params']]
        saved = [p.detach().clone() for p in params]
        had = [('momentum_buffer' in self.optimizer.state.get(p, {})
                and self.optimizer.state[p]['momentum_buffer'] is not None) for p in params]
        saved_buf = [self.optimizer.state[p]['momentum_buffer'].clone() if h else None for p, h in zip(params, had)]
        self._sx, self._sy = inputs.clone(), values.clone()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    self.optimizer.zero_grad(set_to_none=True)
                    self.criterion(self.model(self._sx), self._sy).backward()
                    self.optimizer.step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                self._sloss = self.criterion(self.model(self._sx), self._sy)
                self._sloss.backward()
                self.optimizer.step()
        finally:
            with torch.no_grad():
                for p, keep, h, buf in zip(params, saved, had, saved_buf):
                    p.copy_(keep)
                    mb = self.optimizer.state.get(p, {}).get('momentum_buffer')
                    if mb is not None:
                        mb.copy_(buf) if h else mb.zero_()  # a zero buffer == no buffer for the next update
        self._graph = graph

    def _fit_graph(self, inputs, values):
        if self._graph is None:
            self._capture(inputs, values)
        self._sx.copy_(inputs)
        self._sy.copy_(values)
        self._graph.replay()
        self._bump()
        return self._sloss.detach().clone()

    def _loader(self):
        if self.optimizer is None:
            raise ValueError('Learning rate is not set!')
        if self.data_loader is None:
            self.data_loader = DataLoader(self.memory, self.batch_size, shuffle=True)
        return self.data_loader

    def _bump(self):
        """Tell device-side consumers of the parameters that they changed: a replayed graph updates them without touching
        the tensors' version counters, so CrowdSim.sarl_action keys its re-upload on this counter as well."""
        self.model._cn_weights_epoch = getattr(self.model, '_cn_weights_epoch', 0) + 1

    def _fit(self, inputs, values):
        self.optimizer.zero_grad()
        loss = self.criterion(self.model(inputs.to(self.device)), values.to(self.device))
        loss.backward()
        self.optimizer.step()
        self._bump()
        return loss.data.item()

    def _fit_device(self, inputs, values):
        """_fit without the per-batch host sync: the loss stays a device scalar.  Full batches on a GPU replay the
        captured step."""
        if inputs.is_cuda and inputs.shape[0] == self.batch_size and not self._graph_failed:
            try:
                return self._fit_graph(inputs, values)
            except Exception as exc:  # capture unsupported in this build: the eager step below is always valid
                logging.warning('SGD step graph capture failed (%s); running eagerly', exc)
                self._graph, self._graph_failed = None, True
        self.optimizer.zero_grad()
        loss = self.criterion(self.model(inputs), values)
        loss.backward()
        self.optimizer.step()
        self._bump()
        return loss.detach()

    def _device_memory(self):
        return (isinstance(self.memory, DeviceReplayMemory) and self.memory.device == torch.device(self.device)
                and len(self.memory) > 0)

    def optimize_epoch(self, num_epochs):
        if self.optimizer is None:
            raise ValueError('Learning rate is not set!')
        if self._device_memory():
            average_epoch_loss = 0
            for epoch in range(num_epochs):
                total = torch.zeros((), device=self.memory.device)
                for inputs, values in self.memory.batches(self.batch_size):
                    total += self._fit_device(inputs, values)
                average_epoch_loss = total.item() / len(self.memory)
                logging.debug('Average loss in epoch %d: %.2E', epoch, average_epoch_loss)
            return average_epoch_loss
        loader = self._loader()
        average_epoch_loss = 0
        for epoch in range(num_epochs):
            epoch_loss = sum(self._fit(inputs, values) for inputs, values in loader)
            average_epoch_loss = epoch_loss / len(self.memory)
            logging.debug('Average loss in epoch %d: %.2E', epoch, average_epoch_loss)
        return average_epoch_loss

    def optimize_batch(self, num_batches):
        if self.optimizer is None:
            raise ValueError('Learning rate is not set!')
        if self._device_memory():
            total = torch.zeros((), device=self.memory.device)
            for _ in range(num_batches):  # a fresh permutation per batch, as trainer.py:57 does
                inputs, values = next(self.memory.batches(self.batch_size, limit=1))
                total += self._fit_device(inputs, values)
            average_loss = total.item() / num_batches
            logging.debug('Average loss : %.2E', average_loss)
            return average_loss
        loader = self._loader()
        losses = 0
        for _ in range(num_batches):
            inputs, values = next(iter(loader))  # a fresh shuffled iterator per batch, as trainer.py:57 does
            losses += self._fit(inputs, values)
        average_loss = losses / num_batches
        logging.debug('Average loss : %.2E', average_loss)
        return average_loss
