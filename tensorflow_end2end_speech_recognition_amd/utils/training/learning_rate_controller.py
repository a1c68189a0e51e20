"""Decay learning rate per epoch -- mirror of utils/training/learning_rate_controller.py:11-65."""


class Controller(object):

    def __init__(self, learning_rate_init, decay_start_epoch, decay_rate, decay_patient_epoch=1,
                 lower_better=True, worst_value=1):
        self.learning_rate_init = learning_rate_init
        self.decay_start_epoch = decay_start_epoch
        self.decay_rate = decay_rate
        self.decay_patient_epoch = decay_patient_epoch
        self.not_improved_epoch = 0
        self.lower_better = lower_better
        self.best_value = worst_value

    def decay_lr(self, learning_rate, epoch, value):
        if not self.lower_better:
            value *= -1
        if epoch < self.decay_start_epoch:
            if value < self.best_value:
                self.best_value = value
            return learning_rate
        if value < self.best_value:
            self.best_value = value
            self.not_improved_epoch = 0
            return learning_rate
        elif self.not_improved_epoch < self.decay_patient_epoch:
            self.not_improved_epoch += 1
            return learning_rate
        else:
            self.not_improved_epoch = 0
            return learning_rate * self.decay_rate
