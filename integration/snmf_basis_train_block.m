function [B_DFT_init, B_Mel_init, A_DFT_init, A_Mel_init] = snmf_basis_train_block(s_full, DC_bin, R, p)
% SNMF_BASIS_TRAIN_BLOCK  The numeric block of run_basis_train.m (lines 58-97: features, exemplar init, the two sparse_nmf
%   solves) as ONE MEX call on an MI355X.  A maintainer replaces those lines of run_basis_train.m by
%       [B_DFT_init, B_Mel_init, A_DFT_init, A_Mel_init] = snmf_basis_train_block(s_full, DC_bin_set(l), R, p);
%   and keeps everything around them (wav assembly :16-57, wavwrite :98, normalisation :113-116, k-means :118-134, save :136).
%   TF_mag, TF_Mel and the exemplar columns are formed in HBM; only s_full goes in and the dictionaries / activations come out.
%   The draws stay in MATLAB, in the reference's order: rng(1); randsample(...) (:80-81), then what both sparse_nmf calls draw
%   after re-seeding, rand('seed', p.random_seed); rand(r, n) (src/sparse_nmf.m:112-114,:133-134 -- the same matrix for both).
if ~isfield(p, 'random_seed'), p.random_seed = 1; end
melmat = mel_matrix(p.fs, p.F_order, p.fftlength, 1, p.fs/2)';
pp = p; pp.DCbin = DC_bin;
m = snmf_dnmf_mex('nframes', length(s_full), pp);
rng('default'); rng(1);
sample_idx = randsample(m, p.cluster_buff * R);
H0 = [];
if p.train_Exemplar == 0 && ~(isfield(p, 'snmf_device_rng') && p.snmf_device_rng)
    if p.random_seed > 0, rand('seed', p.random_seed); end %#ok<RAND>
    H0 = rand(p.cluster_buff * R, m);
end
[B_DFT_init, B_Mel_init, A_DFT_init, A_Mel_init] = snmf_dnmf_mex('train', double(s_full(:)), double(sample_idx(:)), H0, p, melmat, DC_bin);
end
