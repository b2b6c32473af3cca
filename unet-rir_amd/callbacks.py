"""The two Keras callbacks the reference's `get_callbacks` returns (dl_models/u_net.py:72-81, res_ae.py:78-87, autoencoder.py:72-81),
restated for `compile_and_fit` of the boundary classes: host-side bookkeeping only."""
import csv


class Callback:
    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


class CSVLogger(Callback):
    """tf.keras.callbacks.CSVLogger(filename, separator=',', append=False): one row per epoch, columns `epoch` + the sorted log keys."""

    def __init__(self, filename, separator=",", append=False):
        self.filename, self.sep, self.append = filename, separator, append
        self._f = self._w = None

    def on_train_begin(self, logs=None):
        self._f = open(self.filename, "a" if self.append else "w", newline="")
        self._w = None

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        keys = sorted(logs)
        if self._w is None:
            self._w = csv.DictWriter(self._f, fieldnames=["epoch"] + keys, delimiter=self.sep)
            if not (self.append and self._f.tell() > 0):
                self._w.writeheader()
        self._w.writerow({"epoch": epoch, **{k: logs[k] for k in keys}})
        self._f.flush()

    def on_train_end(self, logs=None):
        if self._f is not None:
            self._f.close()
            self._f = None


class EarlyStopping(Callback):
    """tf.keras.callbacks.EarlyStopping(monitor, patience) with Keras' defaults (min_delta 0, mode 'min' for a loss, no baseline, weights
    not restored): training stops after `patience` epochs without a new minimum of the monitored value."""

    def __init__(self, monitor="val_loss", min_delta=0.0, patience=0):
        self.monitor, self.min_delta, self.patience = monitor, abs(float(min_delta)), int(patience)

    def on_train_begin(self, logs=None):
        self.wait, self.stopped_epoch, self.best, self.best_epoch = 0, 0, float("inf"), 0

    def on_epoch_end(self, epoch, logs=None):
        current = (logs or {}).get(self.monitor)
        if current is None:
            return
        self.wait += 1
        if current < self.best - self.min_delta:
            self.best, self.best_epoch, self.wait = current, epoch, 0
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.model.stop_training = True
