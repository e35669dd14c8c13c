"""Immediate callers of the hot path, API-compatible with the reference's experiments_methods.py
(same function names and keyword arguments), driving the HIP engine instead of Keras.

    prepare_model_input         ref :12-16    tensors from sessions (FullModelPreprocessor x 2)
    run_model                   ref :19-50    Adagrad(lr, 1e-8, 0, clipnorm=1.) + callbacks + fit
    run_model_with_generator    ref :53-78    one pulled batch per epoch step
    evaluate_full_model         ref :80-88
    analyze_history             ref :91-97    best epoch by validation loss
    run_multinomial, run_markov ref :100-127  count baselines
    run_model_no_recurrence     ref :130-185  NoRecurrenceModel variants
    run_model_with_recurrence   ref :188-249  RNNFullModel variants (every flag set)

Internals are organised around three helpers (_optimizer, _train, _score_saved) rather than the
reference's copy-per-function layout; behaviour per call site is documented next to each helper.
"""
import numpy as np

from . import utils
from .keras_compat import Adagrad, EarlyStopping, ModelCheckpoint
from .model import (MarkovModel, ModelResults, MultinomialModel, NoRecurrenceModel, RNNBaseline, RNNFullModel,
                    ValLossHistoryCut)
from .preprocessor import BaselinePreprocessor, FullModelPreprocessor

# ref :30 writes "<name>.{epoch:02d}-{val_loss:.2f}.hdf5" through h5py; h5py is absent here and the payload is a numpy
# .npz container (keys weight0..N), so the file is NAMED for what it holds -- the reference's load_weights cannot read it
CHECKPOINT_TEMPLATE = ".{epoch:02d}-{val_loss:.2f}.npz"
PATIENCE = 15                                                  # ref :37


def _optimizer(lr):
    """The one optimizer the reference ever builds (ref :41, :70, :162, :218)."""
    return Adagrad(lr=lr, epsilon=1e-08, decay=0.0, clipnorm=1.)


def _callback_list(model, dir_save, want_checkpoint, want_early_stop, cut_metric):
    """ModelCheckpoint(best only, weights only) and EarlyStopping as the reference wires them."""
    out = []
    if want_checkpoint:
        out.append(ModelCheckpoint(dir_save + model.model_name + CHECKPOINT_TEMPLATE, monitor="val_loss",
                                   save_weights_only=True, save_best_only=True))
    if want_early_stop:
        out.append(EarlyStopping(monitor="my_loss" if cut_metric else "val_loss", min_delta=0, patience=PATIENCE,
                                 verbose=0, mode="auto"))
    return out


def _score_saved(model, read_file, lr, batch_size, train, y_train, validation, y_val, loss="categorical_crossentropy"):
    """The `read_file` branch (ref :161-170, :217-228): load weights, report train / val loss."""
    model.compile_model(loss=loss, metrics=[], optimizer=_optimizer(lr))
    model.load_model_weights(read_file)
    res = ModelResults()
    res.train_loss = model.evaluate(train, y_train, batch_size=batch_size)[1][0]
    res.val_loss = model.evaluate(validation, y_val, batch_size=batch_size)[1][0]
    print("train loss: %f, val loss: %f" % (res.train_loss, res.val_loss))
    return res


def prepare_model_input(seqs_train, seqs_val, xs_train, xs_val, vocab, max_seq_length, sparse=False):
    pre = FullModelPreprocessor(vocab=vocab, pad_value=0., seq_length=max_seq_length, sparse=sparse)
    return pre.transform_data(seqs_train, xs=xs_train) + pre.transform_data(seqs_val, xs=xs_val)


def run_model(model, x_train, y_train,
              validation_data=None,
              orig_seqs_lengths=None,
              model_checkpoint=False,
              loss="categorical_crossentropy",
              n_epochs=20,
              batch_size=10,
              verbose=1,
              dir_save="trained_models/",
              early_stopping=False,
              wrt_time=False,
              lr=0.01):
    """Compile with the reference's Adagrad and fit.  With ``wrt_time`` the validation metric is the
    last-30 % NLL of ValLossHistoryCut (and Keras' own validation pass is switched off)."""
    callbacks = []
    cut = None
    if wrt_time:
        cut = ValLossHistoryCut(validation_data, orig_seqs_lengths)
        callbacks.append(cut)
    callbacks += _callback_list(model, dir_save, model_checkpoint, early_stopping, wrt_time)
    model.compile_model(loss=loss, metrics=[], optimizer=_optimizer(lr))
    history = model.fit_model(x_train, y_train, validation_data=None if wrt_time else validation_data,
                              n_epochs=n_epochs, batch_size=batch_size, verbose=verbose, callbacks=callbacks)
    if cut is not None and history:
        history.history["val_loss"] = cut.val_lossses
    return history


def run_model_with_generator(model, train_gen, val_gen,
                             model_checkpoint=False,
                             loss="categorical_crossentropy",
                             n_epochs=20,
                             batch_size=10,
                             verbose=1,
                             dir_save="trained_models/",
                             early_stopping=False,
                             wrt_time=False,
                             lr=0.01,
                             orig_seqs_lengths=None):      # accepted (the reference's caller passes it, ref :238-241)
    callbacks = _callback_list(model, dir_save, model_checkpoint, early_stopping, False)
    model.compile_model(loss=loss, metrics=[], optimizer=_optimizer(lr))
    return model.fit_generator(train_gen, steps_per_epoch=1, validation_data=val_gen, validation_steps=1,
                               epochs=n_epochs, verbose=verbose, callbacks=callbacks)


def evaluate_full_model(model, seqs_test, xs_test, vocab, max_seq_length, with_xs=True):
    pre = FullModelPreprocessor(vocab=vocab, pad_value=0., seq_length=max_seq_length)
    x_test, y_test, test_xs = pre.transform_data(seqs_test, xs=xs_test)
    return model.model.evaluate([x_test, test_xs] if with_xs else x_test, y_test)


def analyze_history(history):
    """ModelResults(train loss at the best epoch, best validation loss, 1-based best epoch)."""
    val = np.asarray(history.history["val_loss"], dtype=np.float64)
    best = int(np.argmin(val))
    train_at_best = history.history["loss"][best]
    print("min val loss: %f at epoch: %d" % (val[best], best + 1))
    print("train loss: %f at epoch: %d" % (train_at_best, best + 1))
    return ModelResults(train_at_best, val[best], best + 1)


def _count_model_results(model, seqs_train, val_seqs, wrt_time, label):
    val_preds = model.predict(val_seqs)
    if wrt_time:
        train_ll, val_ll = utils.compute_likelihood_cut(val_preds, 0.7, count_first_prob=False)
    else:
        train_ll = utils.compute_likelihood(model.predict(seqs_train), count_first_prob=False)
        val_ll = utils.compute_likelihood(val_preds, count_first_prob=False)
    print("%s train neg ll: %f, val neg ll: %f" % (label, train_ll, val_ll))
    return ModelResults(train_ll, val_ll, None)


def run_multinomial(seqs_train, val_seqs, n_classes, wrt_time=False, normalize=True, k=1.0):
    unigram = MultinomialModel(n_classes, model_name="multinomial", k=k)
    unigram.fit_model(seqs_train, normalize=normalize)
    return unigram, _count_model_results(unigram, seqs_train, val_seqs, wrt_time, "Multinomial")


def run_markov(seqs_train, val_seqs, n_classes, wrt_time=False, k=1.0):
    chain = MarkovModel(n_classes, model_name="markov", order=1, k=k)
    chain.fit_model(seqs_train)
    return chain, _count_model_results(chain, seqs_train, val_seqs, wrt_time, "Markov")


def _train(model, train, y_train, validation, y_val, train_gen, val_gen, **kw):
    """Shared tail of the two run_model_* drivers: generator path or array path, then the best epoch."""
    if train_gen is not None and val_gen is not None:
        history = run_model_with_generator(model, train_gen, val_gen, **kw)
    else:
        history = run_model(model, train, y_train, validation_data=(validation, y_val), **kw)
    return analyze_history(history)


def _shapes(x_train, timesteps, x_dim, y_dim):
    return (x_train.shape[1] if timesteps is None else timesteps,
            x_train.shape[2] if x_dim is None else x_dim,
            x_train.shape[2] if y_dim is None else y_dim)


def run_model_no_recurrence(x_train, y_train, train_xs, x_val, y_val, val_xs, vocab,
                            timesteps=None, x_dim=None, y_dim=None,
                            train_gen=None, val_gen=None,
                            orig_seqs_lengths=None, wrt_time=False,
                            read_file=None, model_checkpoint=False, early_stopping=False,
                            n_epochs=50, batch_size=10,
                            y_to_y_trainable=True, y_to_y_w_initializer=None, y_to_y_regularizer=None,
                            verbose=1, model_name="test_model",
                            connect_x=True, connect_y=True, mask_value=0.0, lr=0.01,
                            xy_bias=False, y_bias=False, embed_y=False, z_dim=10, diag_b=True,
                            loss="categorical_crossentropy"):
    timesteps, x_dim, y_dim = _shapes(x_train, timesteps, x_dim, y_dim)
    model = NoRecurrenceModel(timesteps=timesteps, x_dim=x_dim, y_dim=y_dim, model_name=model_name,
                              y_to_y_w_initializer=y_to_y_w_initializer, y_to_y_regularizer=y_to_y_regularizer,
                              y_bias=y_bias, xy_bias=xy_bias, connect_x=connect_x, connect_y=connect_y,
                              embed_y=embed_y, z_dim=z_dim, diag_b=diag_b, mask_value=mask_value)
    train = ([x_train] if connect_y else []) + ([train_xs] if connect_x else [])
    validation = ([x_val] if connect_y else []) + ([val_xs] if connect_x else [])
    if read_file:
        return model, _score_saved(model, read_file, lr, batch_size, train, y_train, validation, y_val, loss)
    if connect_y:
        model.set_layer_weights_trainable("y_output", trainable=y_to_y_trainable)
    results = _train(model, train, y_train, validation, y_val, train_gen, val_gen,
                     orig_seqs_lengths=orig_seqs_lengths, model_checkpoint=model_checkpoint, n_epochs=n_epochs,
                     batch_size=batch_size, verbose=verbose, early_stopping=early_stopping, wrt_time=wrt_time,
                     loss=loss, lr=lr)
    return model, results


def run_model_with_recurrence(x_train, y_train, train_xs, x_val, y_val, val_xs, vocab,
                              timesteps=None, x_dim=None, y_dim=None,
                              train_gen=None, val_gen=None,
                              orig_seqs_lengths=None, wrt_time=False,
                              y_to_y_trainable=True, y_to_y_w_initializer=None,
                              toy_regularizer=None, z_to_z_activation="relu", y_to_y_regularizer=None,
                              read_file=None, rnn_type="LSTM", early_stopping=False,
                              n_epochs=50, model_checkpoint=True, loss="categorical_crossentropy",
                              y_to_z_initializer="glorot_normal",
                              batch_size=10, verbose=1, z_dim=10, model_name="test_model",
                              y_to_z=True, y_to_y=True, x_to_y=True, x_to_z=False, diag_b=True,
                              z_to_y_drop=0.0, y_to_z_dropout=0.0, z_to_z_dropout=0.0, lr=0.01):
    timesteps, x_dim, y_dim = _shapes(x_train, timesteps, x_dim, y_dim)
    wiring = dict(y_to_z=y_to_z, y_to_y=y_to_y, x_to_y=x_to_y, x_to_z=x_to_z, diag_b=diag_b)
    dropouts = dict(z_to_y_dropout=z_to_y_drop, y_to_z_dropout=y_to_z_dropout, z_to_z_dropout=z_to_z_dropout)
    model = RNNFullModel(timesteps=timesteps, x_dim=x_dim, y_dim=y_dim, z_dim=z_dim, model_name=model_name,
                         rnn_type=rnn_type, z_to_z_activation=z_to_z_activation, y_to_y_activation="linear",
                         ytoy_bias=False, z_bias=True, toy_bias=False, toy_regularizer=toy_regularizer,
                         y_to_z_initializer=y_to_z_initializer, y_to_y_regularizer=y_to_y_regularizer,
                         y_to_y_w_initializer=y_to_y_w_initializer, **wiring, **dropouts)
    uses_y, uses_x = (y_to_y or y_to_z), (x_to_y or x_to_z)
    train = ([x_train] if uses_y else []) + ([train_xs] if uses_x else [])
    validation = ([x_val] if uses_y else []) + ([val_xs] if uses_x else [])
    if read_file:
        return model, _score_saved(model, read_file, lr, batch_size, train, y_train, validation, y_val)
    if y_to_y:
        model.set_layer_weights_trainable("y_to_y_output", trainable=y_to_y_trainable)
    results = _train(model, train, y_train, validation, y_val, train_gen, val_gen,
                     orig_seqs_lengths=orig_seqs_lengths, model_checkpoint=model_checkpoint, n_epochs=n_epochs,
                     batch_size=batch_size, verbose=verbose, early_stopping=early_stopping, wrt_time=wrt_time,
                     loss=loss, lr=lr)
    return model, results
