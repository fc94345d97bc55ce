function [B_hat] = run_basis_DNMF(x, d, B, p)
% RUN_BASIS_DNMF  Drop-in replacement of run_basis_DNMF.m of lordet01/SE_SNMF_NAT on an MI355X (libsnmf_hip.so).
%
%   Same signature as the reference.  The whole function -- truncation to equal length, y = x + d, the three spectrogram
%   feature sets and the three sparse_nmf solves (H-only on Y, W-only on X and on D with the activations of solve 1) -- is ONE
%   MEX call (integration/snmf_dnmf_mex.cpp -> snmf_run_basis_dnmf_audio_f64): only the two waveforms and B go to the device,
%   only B_hat comes back; Y, X, D and A_hat never leave HBM.
%
%   What stays here is the one thing that must come from MATLAB for RNG parity with the reference: the initial activations
%   of solve 1.  src/sparse_nmf.m draws them as rand('seed', p.random_seed); h = rand(r, n) (init_w is given, so this is the
%   first draw after the re-seed); the same two statements run below.  Set p.snmf_device_rng = 1 to let the engine draw
%   them on the device instead (no r x n array crosses PCIe; the values then come from Philox, not from MATLAB's generator).
if ~isfield(p, 'random_seed'), p.random_seed = 1; end
n = snmf_dnmf_mex('nframes', min(length(x), length(d)), p);
if isfield(p, 'snmf_device_rng') && p.snmf_device_rng
    H0 = [];
else
    if p.random_seed > 0, rand('seed', p.random_seed); end %#ok<RAND>
    H0 = rand(p.R_x + p.R_d, n);
end
if isfield(p, 'snmf_devices') && numel(p.snmf_devices) > 1
    % BASELINE config 4: the frames of all three solves sharded over the GPUs of this node (zero-based device ordinals), one
    % MEX call (snmf_run_basis_dnmf_multi_f64): the three feature sets are formed on device 0 (snmf_frontend_mex), every rank
    % then takes its shard of each through its own PCIe link, A_hat stays in HBM between the solves, only B_hat comes back.
    m = min(length(x), length(d));
    x = double(x(1:m)); d = double(d(1:m));
    Y = snmf_frontend_mex('stft', x(:) + d(:), p, p.DCbin);
    X = snmf_frontend_mex('stft', x(:), p, p.DCbin);
    D = snmf_frontend_mex('stft', d(:), p, p.DCbin);
    B_hat = snmf_dnmf_mex('dnmf_multi', Y, X, D, double(B), H0, p, double(p.snmf_devices(:)'));
    return
end
B_hat = snmf_dnmf_mex('dnmf', double(x(:)), double(d(:)), double(B), H0, p, []);
end
