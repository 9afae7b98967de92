class TensorShapeProto:
    def __init__(self, *a, **k):
        pass
