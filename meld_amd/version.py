# meld_amd tracks the behaviour of KrishnaswamyLab/MELD 1.0.2 (reference meld/version.py:3)
__version__ = "1.0.2+mi355x.r1"
