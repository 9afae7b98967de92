from . import event_file_loader  # noqa
