"""Entry point mirroring the reference's ``scripts/run_metrics_dnn_classif.py`` for the classifier outputs: reads the
``*_y_hat_hard.pt`` files written by ``scripts/evaluate_*_net.py`` (and the labels saved beside them), computes
accuracy / precision / recall / F1 per utterance and prints the reference's METRIC / AVERAGE / CONF. INT. table.
Run from the package root: ``python scripts/run_metrics_dnn_classif.py [eval_out]``."""
import sys
sys.path.append('.')

from avvad.train import metrics_main

confidence = 0.95  # confidence interval (name as in the reference script)
eps = 1e-8

if __name__ == '__main__':
    metrics_main(sys.argv[1] if len(sys.argv) > 1 else "eval_out", confidence=confidence, eps=eps)
