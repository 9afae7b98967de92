from torch import nn


class ExactGP(nn.Module):
    def __init__(self, train_inputs, train_targets, likelihood):
        super().__init__()
        self.train_inputs = train_inputs
        self.train_targets = train_targets
        self.likelihood = likelihood

    def set_train_data(self, inputs=None, targets=None, strict=True):
        if inputs is not None:
            self.train_inputs = tuple(inputs) if not hasattr(inputs, 'shape') else (inputs,)
        if targets is not None:
            self.train_targets = targets
