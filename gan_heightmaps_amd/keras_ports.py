"""``ReduceLROnPlateau`` with the reference's constructor and callback surface (/root/reference/keras_ports.py:7-111):
a host-side schedule over the shared learning-rate scalar that the optimiser kernels read from HBM
(``updates.shared`` -> ``GanStep.set_lr``).  Checked call for call against the reference's class, executed
(tests/test_reference_trainloop.py).

Behaviour kept from the reference port: there is no monitored-quantity name, so ``mode='auto'`` takes the
"larger is better" branch exactly like ``mode='max'`` (keras_ports.py:68-73); only ``mode='min'`` minimises."""
import numpy as np


class ReduceLROnPlateau:
    def __init__(self, learning_rate, factor=0.1, patience=10, verbose=0, mode='auto', epsilon=1e-4, cooldown=0,
                 min_lr=0):
        if factor >= 1.0:
            raise ValueError('ReduceLROnPlateau does not support a factor >= 1.0.')
        self.learning_rate = learning_rate          # a shared scalar: get_value() / set_value()
        self.factor, self.patience, self.verbose = factor, patience, verbose
        self.mode, self.epsilon, self.cooldown, self.min_lr = mode, epsilon, cooldown, min_lr
        self._reset()

    def _reset(self):
        if self.mode not in ('auto', 'min', 'max'):
            self.mode = 'auto'
        minimise = self.mode == 'min'
        self.monitor_op = (lambda a, b: a < b - self.epsilon) if minimise else (lambda a, b: a > b + self.epsilon)
        self.best = np.inf if minimise else -np.inf
        self.cooldown_counter = 0
        self.wait = 0
        self.lr_epsilon = self.min_lr * 1e-4

    def on_train_begin(self, logs=None):
        self._reset()

    def in_cooldown(self):
        return self.cooldown_counter > 0

    def on_epoch_end(self, monitor, epoch, logs=None):
        if monitor is None:
            return
        if self.in_cooldown():
            self.cooldown_counter -= 1
            self.wait = 0
        if self.monitor_op(monitor, self.best):
            self.best = monitor
            self.wait = 0
        elif not self.in_cooldown():
            if self.wait >= self.patience:
                old = float(self.learning_rate.get_value())
                if old > self.min_lr + self.lr_epsilon:
                    new = max(old * self.factor, self.min_lr)
                    self.learning_rate.set_value(new)
                    if self.verbose > 0:
                        print('\nEpoch %05d: reducing learning rate to %s.' % (epoch, new))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0
            self.wait += 1
