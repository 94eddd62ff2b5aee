"""Learning-rate schedule used by optimizer_factory (mirror of the reference's model/learningrate.py)."""
import torch


class LearningRateSchedule:
    def get_learning_rate(self, epoch):
        raise NotImplementedError


class StepLearningRateSchedule(LearningRateSchedule):
    """lr = initial * factor ** (epoch // interval)   (reference model/learningrate.py:17-25)."""

    def __init__(self, specs):
        self.initial = specs["initial"]
        self.interval = specs["interval"]
        self.factor = specs["factor"]

    def get_learning_rate(self, epoch):
        return self.initial * (self.factor ** (epoch // self.interval))


def adjust_learning_rate(lr_schedules, optimizer, epoch):
    """reference model/learningrate.py:28-34."""
    for i, group in enumerate(optimizer.param_groups):
        sched = lr_schedules[i] if isinstance(lr_schedules, list) else lr_schedules
        lr = sched.get_learning_rate(epoch)
        if torch.is_tensor(group["lr"]):      # capturable optimizers keep the rate on the device (graph replay reads it there)
            group["lr"].fill_(float(lr))
        else:
            group["lr"] = lr


def get_learning_rates(optimizer):
    return [g["lr"] for g in optimizer.param_groups]
