class DictConfig:  # isinstance check only (detectron2/config.py:872)
    pass
