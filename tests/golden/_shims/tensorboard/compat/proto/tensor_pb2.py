class TensorProto:
    def __init__(self, *a, **k):
        pass
