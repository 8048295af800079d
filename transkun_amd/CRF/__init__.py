# mirrors /root/reference/transkun/CRF/__init__.py:1
from .NeuralSemiCRFInterval import *  # noqa: F401,F403
from .NeuralSemiCRFInterval import (NeuralSemiCRFInterval, viterbi, viterbiBackward, computeLogZ,  # noqa: F401
                                    forward_backward, evalPath, computeLogZFasterGrad,
                                    ComputeLogZFasterGrad)
