"""GPU parity for the rows either side of the hot path and for the
reference behaviours around reset(): the same scenarios as the CPU tests, driven through the C ABI
on HipBackend and checked against the oracle sessions / the fixtures of the real reference.

* multi-session scheduler   speechcatcher/speechcatcher_server.py:331-371 (pool), :270 (per-session call)
* server sessions           speechcatcher/speechcatcher_server.py:205-328,359-397
* CLI segment loop          speechcatcher/speechcatcher.py:414-497,574-644
* reset() quirk             speechcatcher/beam_search/scorers.py:342-350
"""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from speechcatcher_amd.hip_backend import HipBackend
    return HipBackend("cuda:0")


def test_hip_reset_keeps_stale_ctc_table_like_the_reference(hip):
    from test_engine_spec import run_reset_quirk
    run_reset_quirk(backend=hip, device="cuda:0", score_tol=1e-3)


@pytest.mark.parametrize("bbd", [0, 1])
def test_hip_calls_after_final_without_reset(hip, bbd):
    from test_engine_spec import run_after_final
    run_after_final(bbd, backend=hip, device="cuda:0", score_tol=1e-3)


def test_hip_scheduler_sessions_with_different_chunking_match_reference(hip):
    from test_scheduler import run_sessions_different_chunking
    run_sessions_different_chunking(backend=hip, device="cuda:0")


def test_hip_segments_serial_strict_equals_reference_cli_semantics(hip):
    from test_scheduler import run_segments_serial_strict
    run_segments_serial_strict(backend=hip, device="cuda:0")


@pytest.mark.parametrize("vosk", [False, True])
def test_hip_server_sessions_equal_private_oracle_sessions(hip, vosk):
    from test_server_session import run_sessions_vs_oracle
    run_sessions_vs_oracle(vosk, backend=hip, device="cuda:0")


def test_hip_strict_reference_server_never_resets(hip):
    from test_server_session import run_strict_server
    run_strict_server(backend=hip, device="cuda:0")


def test_hip_recognize_recording_parallel_streams(hip):
    from test_segmenter import run_recording
    run_recording(2, seconds=70, backend=hip, device="cuda:0")


def test_hip_recognize_recording_serial_equals_reference_cli_semantics(hip):
    from test_segmenter import run_recording
    run_recording(1, seconds=64, backend=hip, device="cuda:0", check_oracle=True)
