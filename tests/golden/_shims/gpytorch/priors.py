class GammaPrior:
    def __init__(self, *a, **k):
        self.args = a
