"""Immediate callers of the hot path: the reference's experiments_methods.py with identical
function names and keyword arguments, running on the HIP engine.

  prepare_model_input         experiments_methods.py:12-16
  run_model                   :19-50   (Adagrad(lr, 1e-8, 0, clipnorm=1.), callbacks, fit)
  run_model_with_generator    :53-78
  evaluate_full_model         :80-88
  analyze_history             :91-97
  run_multinomial/run_markov  :100-127
  run_model_no_recurrence     :130-185 (NoRecurrenceModel: the same logit terms, no recurrent state)
  run_model_with_recurrence   :188-249 (RNNFullModel, every y_to_z / y_to_y / x_to_y / x_to_z flag set)
"""
import numpy as np

from . import utils
from .keras_compat import Adagrad, EarlyStopping, ModelCheckpoint
from .model import (RNNBaseline, RNNFullModel, NoRecurrenceModel, ValLossHistoryCut, MultinomialModel, MarkovModel,
                    ModelResults)
from .preprocessor import FullModelPreprocessor, BaselinePreprocessor


def prepare_model_input(seqs_train, seqs_val, xs_train, xs_val, vocab, max_seq_length, sparse=False):
    pre = FullModelPreprocessor(vocab=vocab, pad_value=0., seq_length=max_seq_length, sparse=sparse)
    x_train, y_train, train_xs = pre.transform_data(seqs_train, xs=xs_train)
    x_val, y_val, val_xs = pre.transform_data(seqs_val, xs=xs_val)
    return x_train, y_train, train_xs, x_val, y_val, val_xs


def _callbacks(model, validation_data, orig_seqs_lengths, model_checkpoint, dir_save, early_stopping, wrt_time):
    cbs, val_history = [], None
    if wrt_time:
        val_history = ValLossHistoryCut(validation_data, orig_seqs_lengths)
        cbs.append(val_history)
    if model_checkpoint:
        cbs.append(ModelCheckpoint(dir_save + model.model_name + ".{epoch:02d}-{val_loss:.2f}.hdf5", monitor="val_loss",
                                   save_weights_only=True, save_best_only=True))
    if early_stopping:
        cbs.append(EarlyStopping(monitor="my_loss" if wrt_time else "val_loss", min_delta=0, patience=15, verbose=0,
                                 mode="auto"))
    return cbs, val_history


def run_model(model, x_train, y_train, validation_data=None, orig_seqs_lengths=None, model_checkpoint=False,
              loss="categorical_crossentropy", n_epochs=20, batch_size=10, verbose=1, dir_save="trained_models/",
              early_stopping=False, wrt_time=False, lr=0.01):
    cbs, val_history = _callbacks(model, validation_data, orig_seqs_lengths, model_checkpoint, dir_save, early_stopping,
                                  wrt_time)
    rnn_validation_data = None if wrt_time else validation_data
    adam = Adagrad(lr=lr, epsilon=1e-08, decay=0.0, clipnorm=1.)      # (named `adam` in the reference too)
    model.compile_model(loss=loss, metrics=[], optimizer=adam)
    history = model.fit_model(x_train, y_train, validation_data=rnn_validation_data, n_epochs=n_epochs,
                              batch_size=batch_size, verbose=verbose, callbacks=cbs)
    if wrt_time and history:
        history.history["val_loss"] = val_history.val_lossses
    return history


def run_model_with_generator(model, train_gen, val_gen, model_checkpoint=False, loss="categorical_crossentropy",
                             n_epochs=20, batch_size=10, verbose=1, dir_save="trained_models/", early_stopping=False,
                             wrt_time=False, lr=0.01, orig_seqs_lengths=None):
    cbs, _ = _callbacks(model, None, None, model_checkpoint, dir_save, early_stopping, False)
    adam = Adagrad(lr=lr, epsilon=1e-08, decay=0.0, clipnorm=1.)
    model.compile_model(loss=loss, metrics=[], optimizer=adam)
    return model.fit_generator(train_gen, steps_per_epoch=1, validation_data=val_gen, validation_steps=1,
                               epochs=n_epochs, verbose=verbose, callbacks=cbs)


def evaluate_full_model(model, seqs_test, xs_test, vocab, max_seq_length, with_xs=True):
    pre = FullModelPreprocessor(vocab=vocab, pad_value=0., seq_length=max_seq_length)
    x_test, y_test, test_xs = pre.transform_data(seqs_test, xs=xs_test)
    if with_xs:
        return model.model.evaluate([x_test, test_xs], y_test)
    return model.model.evaluate(x_test, y_test)


def analyze_history(history):
    train_losses = history.history["loss"]
    val_losses = history.history["val_loss"]
    best = int(np.argmin(val_losses))
    print("min val loss: %f at epoch: %d" % (np.min(val_losses), best + 1))
    print("train loss: %f at epoch: %d" % (train_losses[best], best + 1))
    return ModelResults(train_losses[best], np.min(val_losses), best + 1)


def _count_model_results(model, seqs_train, val_seqs, wrt_time, label):
    val_preds = model.predict(val_seqs)
    if wrt_time:
        train_neg_ll, val_neg_ll = utils.compute_likelihood_cut(val_preds, 0.7, count_first_prob=False)
    else:
        train_neg_ll = utils.compute_likelihood(model.predict(seqs_train), count_first_prob=False)
        val_neg_ll = utils.compute_likelihood(val_preds, count_first_prob=False)
    print("%s train neg ll: %f, val neg ll: %f" % (label, train_neg_ll, val_neg_ll))
    return ModelResults(train_neg_ll, val_neg_ll, None)


def run_multinomial(seqs_train, val_seqs, n_classes, wrt_time=False, normalize=True, k=1.0):
    model = MultinomialModel(n_classes, model_name="multinomial", k=k)
    model.fit_model(seqs_train, normalize=normalize)
    return model, _count_model_results(model, seqs_train, val_seqs, wrt_time, "Multinomial")


def run_markov(seqs_train, val_seqs, n_classes, wrt_time=False, k=1.0):
    model = MarkovModel(n_classes, model_name="markov", order=1, k=k)
    model.fit_model(seqs_train)
    return model, _count_model_results(model, seqs_train, val_seqs, wrt_time, "Markov")


def run_model_no_recurrence(x_train, y_train, train_xs, x_val, y_val, val_xs, vocab, timesteps=None, x_dim=None, y_dim=None,
                            train_gen=None, val_gen=None, orig_seqs_lengths=None, wrt_time=False, read_file=None,
                            model_checkpoint=False, early_stopping=False, n_epochs=50, batch_size=10, y_to_y_trainable=True,
                            y_to_y_w_initializer=None, y_to_y_regularizer=None, verbose=1, model_name="test_model",
                            connect_x=True, connect_y=True, mask_value=0.0, lr=0.01, xy_bias=False, y_bias=False,
                            embed_y=False, z_dim=10, diag_b=True, loss="categorical_crossentropy"):
    """experiments_methods.py:130-185."""
    timesteps = x_train.shape[1] if timesteps is None else timesteps
    x_dim = x_train.shape[2] if x_dim is None else x_dim
    y_dim = x_train.shape[2] if y_dim is None else y_dim
    model = NoRecurrenceModel(timesteps=timesteps, x_dim=x_dim, y_dim=y_dim, model_name=model_name,
                              y_to_y_w_initializer=y_to_y_w_initializer, y_bias=y_bias, xy_bias=xy_bias, connect_x=connect_x,
                              y_to_y_regularizer=y_to_y_regularizer, connect_y=connect_y, embed_y=embed_y, z_dim=z_dim,
                              diag_b=diag_b, mask_value=mask_value)
    if connect_x and connect_y:
        train, validation = [x_train, train_xs], [x_val, val_xs]
    elif connect_x:
        train, validation = [train_xs], [val_xs]
    else:
        train, validation = [x_train], [x_val]
    if read_file:
        model.compile_model(loss=loss, metrics=[], optimizer=Adagrad(lr=lr, epsilon=1e-08, decay=0.0, clipnorm=1.))
        model.load_model_weights(read_file)
        results = ModelResults()
        results.train_loss = model.evaluate(train, y_train, batch_size=batch_size)[1][0]
        results.val_loss = model.evaluate(validation, y_val, batch_size=batch_size)[1][0]
        return model, results
    if connect_y:
        model.set_layer_weights_trainable("y_output", trainable=y_to_y_trainable)
    if train_gen is not None and val_gen is not None:
        history = run_model_with_generator(model, train_gen, val_gen, model_checkpoint=model_checkpoint, n_epochs=n_epochs,
                                           batch_size=batch_size, verbose=verbose, early_stopping=early_stopping,
                                           wrt_time=wrt_time, loss=loss, lr=lr)
    else:
        history = run_model(model, train, y_train, validation_data=(validation, y_val), orig_seqs_lengths=orig_seqs_lengths,
                            model_checkpoint=model_checkpoint, n_epochs=n_epochs, batch_size=batch_size, verbose=verbose,
                            early_stopping=early_stopping, wrt_time=wrt_time, loss=loss, lr=lr)
    return model, analyze_history(history)


def run_model_with_recurrence(x_train, y_train, train_xs, x_val, y_val, val_xs, vocab, timesteps=None, x_dim=None,
                              y_dim=None, train_gen=None, val_gen=None, orig_seqs_lengths=None, wrt_time=False,
                              y_to_y_trainable=True, y_to_y_w_initializer=None, toy_regularizer=None,
                              z_to_z_activation="relu", y_to_y_regularizer=None, read_file=None, rnn_type="LSTM",
                              early_stopping=False, n_epochs=50, model_checkpoint=True, loss="categorical_crossentropy",
                              y_to_z_initializer="glorot_normal", batch_size=10, verbose=1, z_dim=10,
                              model_name="test_model", y_to_z=True, y_to_y=True, x_to_y=True, x_to_z=False, diag_b=True,
                              z_to_y_drop=0.0, y_to_z_dropout=0.0, z_to_z_dropout=0.0, lr=0.01):
    timesteps = x_train.shape[1] if timesteps is None else timesteps
    x_dim = x_train.shape[2] if x_dim is None else x_dim
    y_dim = x_train.shape[2] if y_dim is None else y_dim
    model = RNNFullModel(timesteps=timesteps, x_dim=x_dim, y_dim=y_dim, z_dim=z_dim, model_name=model_name,
                         rnn_type=rnn_type, z_to_z_activation=z_to_z_activation, y_to_y_activation="linear",
                         ytoy_bias=False, z_bias=True, toy_bias=False, toy_regularizer=toy_regularizer,
                         y_to_z_initializer=y_to_z_initializer, y_to_y_regularizer=y_to_y_regularizer,
                         y_to_y_w_initializer=y_to_y_w_initializer, y_to_z=y_to_z, y_to_y=y_to_y, x_to_y=x_to_y,
                         x_to_z=x_to_z, diag_b=diag_b, z_to_y_dropout=z_to_y_drop, y_to_z_dropout=y_to_z_dropout,
                         z_to_z_dropout=z_to_z_dropout)
    train, validation = [], []
    if y_to_y or y_to_z:
        train.append(x_train)
        validation.append(x_val)
    if x_to_y or x_to_z:
        train.append(train_xs)
        validation.append(val_xs)
    if read_file:
        adam = Adagrad(lr=lr, epsilon=1e-08, decay=0.0, clipnorm=1.)
        model.compile_model(loss="categorical_crossentropy", metrics=[], optimizer=adam)
        model.load_model_weights(read_file)
        results = ModelResults()
        _, scores = model.evaluate(train, y_train, batch_size=batch_size)
        results.train_loss = scores[0]
        _, scores = model.evaluate(validation, y_val, batch_size=batch_size)
        results.val_loss = scores[0]
        print("train loss: %f, val loss: %f" % (results.train_loss, results.val_loss))
        return model, results
    if y_to_y:
        model.set_layer_weights_trainable("y_to_y_output", trainable=y_to_y_trainable)
    if train_gen is not None and val_gen is not None:
        history = run_model_with_generator(model, train_gen, val_gen, orig_seqs_lengths=orig_seqs_lengths,
                                           model_checkpoint=model_checkpoint, n_epochs=n_epochs, batch_size=batch_size,
                                           verbose=verbose, early_stopping=early_stopping, wrt_time=wrt_time, loss=loss,
                                           lr=lr)
    else:
        history = run_model(model, train, y_train, validation_data=(validation, y_val),
                            orig_seqs_lengths=orig_seqs_lengths, model_checkpoint=model_checkpoint, n_epochs=n_epochs,
                            batch_size=batch_size, verbose=verbose, early_stopping=early_stopping, wrt_time=wrt_time,
                            loss=loss, lr=lr)
    return model, analyze_history(history)
