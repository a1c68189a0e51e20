"""Per-epoch learning-rate decay on a stalled validation score.

Same constructor / decay_lr(learning_rate, epoch, value) contract and attributes as the reference's
utils/training/learning_rate_controller.py:11-65 Controller; pinned to 24 trajectories recorded from it
(tests/golden/controller_v1.json).  Rule: scores are compared as "lower is better" (negated otherwise); the best
score is tracked from the first epoch; from `decay_start_epoch` on, an epoch that does not beat the best uses up one
unit of patience, and once `decay_patient_epoch` of them have accumulated the next such epoch multiplies the rate
by `decay_rate` and resets the count."""


class Controller(object):

    def __init__(self, learning_rate_init, decay_start_epoch, decay_rate, decay_patient_epoch=1,
                 lower_better=True, worst_value=1):
        self.learning_rate_init, self.decay_rate = learning_rate_init, decay_rate
        self.decay_start_epoch, self.decay_patient_epoch = decay_start_epoch, decay_patient_epoch
        self.lower_better = lower_better
        self.best_value = worst_value
        self.not_improved_epoch = 0

    def decay_lr(self, learning_rate, epoch, value):
        score = value if self.lower_better else -value
        improved = score < self.best_value
        if improved:
            self.best_value = score
        if epoch < self.decay_start_epoch:
            return learning_rate                      # warm-up: only the best score is tracked
        if improved:
            self.not_improved_epoch = 0
            return learning_rate
        if self.not_improved_epoch >= self.decay_patient_epoch:
            self.not_improved_epoch = 0
            return learning_rate * self.decay_rate
        self.not_improved_epoch += 1
        return learning_rate
