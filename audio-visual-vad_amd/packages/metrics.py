"""Drop-in for the reporting helpers of ``packages/metrics.py`` that the classifier pipeline uses
(``mean_confidence_interval`` ``:5-10``, ``compute_stats`` ``:62-130``; called from
``scripts/run_metrics_dnn_classif.py:292-300``).  Host-side reporting: Student-t confidence half-width of
the per-utterance scores, overall and per input SNR / noise type / speaker, printed as the reference's
``METRIC / AVERAGE / CONF. INT.`` tables (and returned, which the reference does not do).

The speech-enhancement metrics of that file (SI-SDR components, energy ratios) belong to a different
pipeline and are not part of this build."""
import numpy as np
import scipy.stats


def mean_confidence_interval(data, confidence=0.95, round=3):
    """(mean, half-width of the two-sided Student-t interval), both rounded to 3 decimals -- the reference
    ignores its ``round`` argument (``:10``) and so does this."""
    a = np.asarray(data, dtype=np.float64)
    n = a.shape[0]
    half = scipy.stats.sem(a) * scipy.stats.t.ppf(0.5 * (1.0 + confidence), n - 1)
    return np.round(a.mean(), 3), np.round(half, 3)


def _table(columns, rows, confidence):
    """columns: {name: per-utterance values}; rows: index array / mask selecting the utterances of this table."""
    print("{:<10} {:<10} {:<10}".format('METRIC', 'AVERAGE', 'CONF. INT.'))
    out = {}
    for name, values in columns.items():
        m, h = mean_confidence_interval(np.asarray(values)[rows], confidence=confidence)
        out[name] = {'avg': m, '+/-': h}
        print("{:<10} {:<10} {:<10}".format(name, m, h))
    print('\n')
    return out


def compute_stats(metrics_keys, all_metrics, model_data_dir, confidence, all_snr_db=None, all_noise_types=None,
                  all_speakers=None):
    """all_metrics: one tuple of scores per utterance, in ``metrics_keys`` order.  Prints the overall table, then one
    table per distinct SNR / noise type / speaker when those per-utterance labels are given (SNRs ascending as in the
    reference; noise types and speakers in sorted order -- the reference iterates a ``set``).  ``model_data_dir`` is
    accepted for signature compatibility (the reference's json dump is commented out, ``:83-85``)."""
    n = len(all_metrics)
    columns = {key: np.array([row[j] for row in all_metrics], dtype=np.float64) for j, key in enumerate(metrics_keys)}
    stats = {'all': _table(columns, np.arange(n), confidence)}
    groups = (('Input SNR = {:.2f}', all_snr_db, 'snr'), ('Noise type = {}', all_noise_types, 'noise'),
              ('Speaker = {}', all_speakers, 'speaker'))
    for title, labels, tag in groups:
        if labels is None:
            continue
        labels = np.asarray(labels)
        for value in np.unique(labels):
            print(title.format(value))
            stats[(tag, value.item() if hasattr(value, 'item') else value)] = _table(columns, labels == value, confidence)
    return stats
