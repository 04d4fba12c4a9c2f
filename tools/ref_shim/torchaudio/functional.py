from speechcatcher_amd.mel import melscale_fbanks_slaney


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    assert norm == "slaney" and mel_scale == "slaney"
    return melscale_fbanks_slaney(n_freqs, f_min, f_max, n_mels, sample_rate)
