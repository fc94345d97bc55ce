function [B_hat] = run_basis_DNMF_Mel(x, d, B, p)
% RUN_BASIS_DNMF_MEL  Drop-in replacement of run_basis_DNMF_Mel.m on an MI355X: as integration/run_basis_DNMF.m, with the
%   three feature sets projected onto the Mel filter bank on the device (the reference's melmat, run_basis_DNMF_Mel.m:21-23).
if ~isfield(p, 'random_seed'), p.random_seed = 1; end
melmat = mel_matrix(p.fs, p.F_order, p.fftlength, 1, p.fs/2)';
n = snmf_dnmf_mex('nframes', min(length(x), length(d)), p);
if isfield(p, 'snmf_device_rng') && p.snmf_device_rng
    H0 = [];
else
    if p.random_seed > 0, rand('seed', p.random_seed); end %#ok<RAND>
    H0 = rand(p.R_x + p.R_d, n);
end
B_hat = snmf_dnmf_mex('dnmf', double(x(:)), double(d(:)), double(B), H0, p, melmat);
end
