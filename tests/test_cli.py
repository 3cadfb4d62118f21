"""tools/test_model.py against the reference's command line (code/test-model.py:45-85): the argument vectors below are
the ones scripts/test-model-loss.sh:57-73 (TOY) and :87-113 (REAL) issue, token for token."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from helpers import ROOT, load, oracle_weights

W_G_WOW = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
W_D_WOW = "DiffDelGRU-HS[64]-L[ESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
W_G_AKAI = "GRU-HS[64]-L[ESR]-DS[ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL]_BEST"
W_D_AKAI = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL]_BEST"


def cli_module(tag="ntm_cli_r4"):
    spec = importlib.util.spec_from_file_location(tag, os.path.join(ROOT, "tools", "test_model.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    return cli


def loss_script_argv(model, weight, dataset, subset, mode):
    """scripts/test-model-loss.sh:57-63 / :67-73 / :87-93 / :107-113 with the shell variables substituted."""
    return ["--MODEL", model, "--WEIGHTS", weight,
            "--DATASET", dataset, "--SUBSET", subset, "--NO_SHUFFLE", "--SEGMENT_LENGTH", str(44100 * 10),
            mode,
            "--COMPUTE_LOSS",
            "--SAVE_AUDIO",
            "--DESCRIPTIVE_NAME", "LOSS"]


SCRIPT_VECTORS = [
    ("GRU", W_G_WOW, "ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER", "--ADD_DELAY"),
    ("GRU", W_G_WOW, "ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER", "--DEMODULATE"),
    ("DiffDelGRU", W_D_WOW, "ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER", "--ADD_DELAY"),
    ("GRU", W_G_AKAI, "ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL", "--ADD_DELAY"),
    ("GRU", W_G_AKAI, "ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL", "--DEMODULATE"),
    ("DiffDelGRU", W_D_AKAI, "ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL", "--ADD_DELAY"),
]


@pytest.mark.parametrize("model,weight,dataset,mode", SCRIPT_VECTORS)
def test_parser_takes_the_loss_script_command_lines(model, weight, dataset, mode):
    cli = cli_module()
    a = cli.parse_args(loss_script_argv(model, weight, dataset, "Test", mode))
    assert a.MODEL == model and a.WEIGHTS == [weight] and a.DATASET == dataset and a.SUBSET == "Test"
    assert a.NO_SHUFFLE and a.COMPUTE_LOSS and a.SAVE_AUDIO and a.DESCRIPTIVE_NAME == "LOSS" and a.SEGMENT_LENGTH == 441000
    assert a.ADD_DELAY == (mode == "--ADD_DELAY") and a.DEMODULATE == (mode == "--DEMODULATE")
    assert cli.dataset_path(a) == os.path.join("../audio/", dataset)           # code/test-model.py:107,147
    cli.check_model_flag(a.MODEL, a.WEIGHTS)
    name, sd = cli.resolve_weights(weight, a.MODEL_PATH)                        # no ../weights/ here: the exported checkpoint
    assert name == weight and sd is None


def test_parser_defaults_and_every_reference_flag():
    """Types and defaults of code/test-model.py:45-85; every flag of the reference parses."""
    cli = cli_module()
    a = cli.parse_args([])
    ref_defaults = dict(DESCRIPTIVE_NAME=None, SAVE_FIG=False, SAVE_AUDIO=False, MODEL="GRU", ADD_DELAY=False, DELAY_TYPE="Real",
                        ADD_NOISE=False, NOISE_TYPE="Real", DATASET="ReelToReel_Dataset_MiniPulse100_CHOWTAPE", SUBSET="Val",
                        FRACTION=1.0, SEGMENT_LENGTH=None, NO_SHUFFLE=False, DEMODULATE=False, IDX=None, SYNC=0.0,
                        COMPUTE_LOSS=False, DATASET_NOISE="Silence_AKAI_IPS[7.5]_MAXELL", PLOT_SWEEP=False,
                        PLOT_TRANSFER=False, PLOT_PHASE=False, PLOT_DELAY=False, ZOOM=None)
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k
    assert a.WEIGHTS == ["GRU-HS[64]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]"]
    with pytest.raises(AttributeError):                                         # no -L[...] field: parse_loss fails, as upstream
        cli.resolve_weights(a.WEIGHTS[0], a.MODEL_PATH)
    full = ["--DESCRIPTIVE_NAME", "X", "--SAVE_FIG", "--SAVE_AUDIO", "--MODEL", "DiffDelGRU", "--WEIGHTS", W_G_WOW, W_D_WOW,
            "--ADD_DELAY", "--DELAY_TYPE", "True", "--ADD_NOISE", "--NOISE_TYPE", "Generated", "--DATASET", "D", "--SUBSET", "Train",
            "--FRACTION", "0.25", "--SEGMENT_LENGTH", "44100", "--NO_SHUFFLE", "--DEMODULATE", "--IDX", "None", "--SYNC", "0.5",
            "--COMPUTE_LOSS", "--DATASET_NOISE", "N", "--PLOT_SWEEP", "--PLOT_TRANSFER", "--PLOT_PHASE", "--ZOOM", "0.1"]
    a = cli.parse_args(full)
    assert a.WEIGHTS == [W_G_WOW, W_D_WOW] and a.IDX is None and a.FRACTION == 0.25 and a.ZOOM == 0.1 and a.DELAY_TYPE == "True"
    assert cli.parse_args(["--IDX", "7"]).IDX == 7
    cli.check_model_flag("DiffDelGRU", a.WEIGHTS)                               # the LAST weight decides (code/test-model.py:345-353)
    with pytest.raises(SystemExit):
        cli.check_model_flag("GRU", a.WEIGHTS)
    with pytest.raises(AssertionError):
        cli.parse_args(["--DELAY_TYPE", "bogus"])
    with pytest.raises(AssertionError):
        cli.parse_args(["--PLOT_SWEEP", "--PLOT_TRANSFER", "--PLOT_DELAY"])
    with pytest.raises(SystemExit):
        cli.resolve_weights("GRU-HS[64]-L[ESR]-DS[NoSuchDataset]_9", "../weights/")


def test_parser_takes_the_other_evaluation_scripts():
    """The command lines of scripts/test-model-prediction.sh:60-76, test-model-sweep.sh:56-62, test-model-hysteresis.sh:63-100 and
    test-model-noise.sh with their shell variables substituted: all parse; the figure / noise flags land in the "ignored" set."""
    cli = cli_module()
    ds = "ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL"
    prediction = ["--MODEL", "DiffDelGRU", "--WEIGHTS", W_D_AKAI, "--DATASET", ds, "--SEGMENT_LENGTH", "441000",
                  "--SYNC", "0.0", "--NO_SHUFFLE", "--IDX", "3", "--ADD_DELAY", "--DELAY_TYPE", "Real", "--SAVE_AUDIO",
                  "--DESCRIPTIVE_NAME", "PREDICTION_3_0_DELAY[Real]"]
    noised = prediction[:-2] + ["--DATASET_NOISE", "Silence_AKAI_IPS[7.5]_MAXELL", "--ADD_NOISE", "--NOISE_TYPE", "Real",
                                "--DESCRIPTIVE_NAME", "PREDICTION_3_0_DELAY[Real]_NOISE[Real]"]
    sweep = ["--MODEL", "GRU", "--WEIGHTS", W_G_AKAI, "--DATASET", "SinesFadedShortContinuousPulse100_AKAI_IPS[7.5]_MAXELL",
             "--SYNC", "5.0", "--SEGMENT_LENGTH", "441000", "--ZOOM", "10.0", "--NO_SHUFFLE", "--IDX", "0", "--DEMODULATE",
             "--PLOT_SWEEP", "--SAVE_FIG", "--DESCRIPTIVE_NAME", "SWEEP_0"]
    hysteresis = ["--MODEL", "DiffDelGRU", "--WEIGHTS", W_D_AKAI, W_D_WOW, "--DATASET", ds, "--SYNC", "0.0", "--SEGMENT_LENGTH", "44100",
                  "--ZOOM", "0.1", "--NO_SHUFFLE", "--IDX", "7", "--DEMODULATE", "--PLOT_TRANSFER", "--SAVE_FIG", "--DESCRIPTIVE_NAME", "HYSTERESIS_7"]
    for argv, ignored in ((prediction, []), (noised, ["ADD_NOISE"]), (sweep, ["SAVE_FIG", "PLOT_SWEEP", "ZOOM"]),
                          (hysteresis, ["SAVE_FIG", "PLOT_TRANSFER", "ZOOM"])):
        a = cli.parse_args(argv)
        cli.check_model_flag(a.MODEL, a.WEIGHTS)
        assert [k for k in cli.OUT_OF_SCOPE if getattr(a, k)] == ignored
        assert a.SUBSET == "Val" and a.NO_SHUFFLE and not a.COMPUTE_LOSS and isinstance(a.IDX, int)
    assert cli.parse_args(hysteresis).WEIGHTS == [W_D_AKAI, W_D_WOW]
    assert cli.parse_args(sweep).SYNC == 5.0 and cli.parse_args(sweep).ZOOM == 10.0


def test_weights_resolution_prefers_best_pth(tmp_path):
    """<MODEL_PATH>/<name>/best.pth (code/test-model.py:198-199,233) wins over the exported checkpoint of that name."""
    import torch
    cli = cli_module()
    d = tmp_path / "weights" / W_G_WOW
    d.mkdir(parents=True)
    sd = {"GRU.weight_ih_l0": torch.zeros(192, 1)}
    torch.save(sd, str(d / "best.pth"))
    name, got = cli.resolve_weights(W_G_WOW, str(tmp_path / "weights"))
    assert name == W_G_WOW and set(got) == set(sd)
    name, got = cli.resolve_weights(str(d), "nowhere")                          # a directory given directly
    assert name == W_G_WOW and set(got) == set(sd)


def test_all_44_checkpoints_are_exported():
    import ntm_amd
    tab = __import__("json").load(open(os.path.join(ROOT, "tests", "golden", "g7_name_parsers.json")))
    assert sorted(r["name"] for r in tab["names"]) == ntm_amd.weights.available() and len(ntm_amd.weights.available()) == 44
    for r in tab["names"]:
        sd = ntm_amd.weights.load_state_dict(r["name"])
        assert sd["GRU.weight_hh_l0"].shape == (192, 64) and ("output.bias" in sd) == (r["model"] == "GRU")


def _wow_dataset(root, name, subset="Test"):
    """A stereo (audio + pilot pulse train) dataset file pair from golden g12's pulse trains."""
    from scipy.io import wavfile
    g = load("g12_delay_analysis.npz")
    fs, N = int(g["fs"]), len(g["in1"])
    d = root / name / subset
    d.mkdir(parents=True)
    rng = np.random.default_rng(4)
    audio = rng.uniform(-0.4, 0.4, N).astype(np.float32)
    tgt_audio = (0.2 * np.roll(audio, 1200)).astype(np.float32)
    wavfile.write(str(d / "input_0_.wav"), fs, np.stack([audio, g["in1"]], 1))
    wavfile.write(str(d / "target_0_.wav"), fs, np.stack([tgt_audio, g["out1"]], 1))
    return g, fs, N, audio, tgt_audio


@pytest.mark.gpu
def test_loss_script_command_runs_end_to_end(tmp_path, monkeypatch):
    """The reference's canonical command (scripts/test-model-loss.sh:57-63) run from a `scripts/`-like directory next to
    `audio/`: --DATASET resolved under ../audio/, losses against the oracle, the cache under .temp/loss/... written and
    re-loaded, the example prediction exported to ../results/ as 16-bit WAV."""
    import oracle
    from scipy.io import wavfile
    cli = cli_module("ntm_cli_r4_gpu")
    ds = "ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER"
    g, fs, N, audio, tgt_audio = _wow_dataset(tmp_path / "audio", ds)
    (tmp_path / "scripts").mkdir()
    monkeypatch.chdir(tmp_path / "scripts")
    L = 14000
    argv = ["--MODEL", "GRU", "--WEIGHTS", W_G_WOW, "--DATASET", ds, "--SUBSET", "Test", "--NO_SHUFFLE", "--SEGMENT_LENGTH", str(L),
            "--ADD_DELAY", "--COMPUTE_LOSS", "--SAVE_AUDIO", "--DESCRIPTIVE_NAME", "LOSS", "--IDX", "1", "--DELAY_TYPE", "True"]
    got = cli.main(argv)
    T = g["T1"]
    Dn = int(1.25 * T.max() * fs)
    init = 1 << (int(T.max() * fs) - 1).bit_length()
    w = oracle_weights(W_G_WOW)
    nseg = N // L
    X = np.stack([audio[k * L:(k + 1) * L] for k in range(nseg)])
    Dt = np.stack([(T[k * L:(k + 1) * L]).astype(np.float32) * np.float32(fs) for k in range(nseg)])
    yo, _ = oracle.gru_predict(w, X)
    yd, _ = oracle.delay_forward(yo, Dt, np.zeros((nseg, Dn), np.float32))
    Tg = np.stack([tgt_audio[k * L:(k + 1) * L] for k in range(nseg)])
    want = float(np.mean(oracle.esr_per_segment(yd, Tg, init)))
    assert abs(got["ESR"] - want) < 1e-3 * want, (got, want)
    cache = tmp_path / "scripts" / ".temp" / "loss" / ds / "Test"
    assert len(list(cache.glob("*.npy"))) == 1
    assert cli.main(argv + ["--NO_EXAMPLE"]) == got                              # second run: "Loading pre-computed!"
    # the example: segment 1, delayed by its own trajectory, cut at the first delay value (code/test-model.py:512-521)
    res = tmp_path / "results"
    fsr, pred = wavfile.read(str(res / f"{ds}_LOSS_prediction_Supervised 1.wav"))
    _, inp = wavfile.read(str(res / f"{ds}_LOSS_input.wav"))
    start = int(Dt[1, 0])
    assert fsr == fs and pred.dtype == np.int16 and len(pred) == L - start == len(inp)
    assert np.abs(pred.astype(np.float64) / 32767 - yd[1, start:]).max() < 1e-5 + 0.5 / 32767
    assert np.abs(inp.astype(np.float64) / 32767 - X[1, start:]).max() < 0.51 / 32767


@pytest.mark.gpu
def test_cli_two_models_fraction_and_diffdel_example(tmp_path, monkeypatch):
    """`--WEIGHTS a b` builds both (code/test-model.py:192-247), the loss runs the LAST one; `--FRACTION` keeps the first
    int(n * fraction) segments with --NO_SHUFFLE (code/dataset.py:295-341); a mismatching --MODEL is refused."""
    import oracle
    cli = cli_module("ntm_cli_r4_gpu2")
    g, fs, N, audio, tgt_audio = _wow_dataset(tmp_path, "Wow")
    monkeypatch.chdir(tmp_path)
    L = 11000
    W_D = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
    argv = ["--MODEL", "DiffDelGRU", "--WEIGHTS", W_G_WOW, W_D, "--DATASET_DIR", str(tmp_path / "Wow"), "--SUBSET", "Test",
            "--NO_SHUFFLE", "--FRACTION", "0.5", "--SEGMENT_LENGTH", str(L), "--COMPUTE_LOSS", "--IDX", "0", "--DELAY_TYPE", "True",
            "--SAVE_AUDIO", "--RESULTS_PATH", str(tmp_path / "res"), "--PLOT_DELAY", "--ZOOM", "0.5"]
    got = cli.main(argv)
    T = g["T1"]
    nseg = int((N // L) * 0.5)
    max_delay_n = int(1.25 * T.max() * fs)
    init = 1 << (int(T.max() * fs) - 1).bit_length()
    X = np.stack([audio[k * L:(k + 1) * L] for k in range(nseg)])
    Dt = np.stack([(T[k * L:(k + 1) * L]).astype(np.float32) * np.float32(fs) for k in range(nseg)])
    yo, _, _, _ = oracle.diffdel_predict(oracle_weights(W_D), X, Dt, max_delay_n)
    Tg = np.stack([tgt_audio[k * L:(k + 1) * L] for k in range(nseg)])
    want = float(np.mean(oracle.esr_per_segment(yo, Tg, init)))
    assert abs(got["ESR"] - want) < 1e-3 * want, (got, want)
    assert sorted(os.listdir(tmp_path / "res")) == ["Wow_input.wav", "Wow_prediction_Supervised 1.wav",
                                                    "Wow_prediction_Supervised 2.wav", "Wow_target.wav"]
    with pytest.raises(SystemExit):
        cli.main(["--MODEL", "GRU"] + argv[2:])
