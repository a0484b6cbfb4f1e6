"""Detector geometry and photometric constants of the L1->L2 path.

Values follow the reference's ``src/romanimpreprocess/pars.py:8-21``; they are
physical facts about the Roman WFI H4RG-10 detectors, not tunables.
"""

# full frame, reference-pixel border and read-out channels
nside = 4096
nborder = 4
nchannel = 32

nside_active = nside - 2 * nborder  # 4088
channelwidth = nside // nchannel  # 128 columns per channel (and of the reference output, "amp33")
nside_augmented = nside + channelwidth  # 4224 = science frame with amp33 appended on the right

# photometry (docs/conventions.tex in the reference)
Omega_ideal = 2.8440360952308436e-13  # (0.11 arcsec)^2 in sr
h_Planck = 6.62607015e-24
g_ideal = 1.458
