% claudio_tracked_hip.m — GPU drop-in for the three tracked-ranging jobs of the reference's acquisition/ directory:
%
%   claudio_tracked_hip('ranging')   ~  acquisition/claudio_aligned_code_ranging_separate.m
%   claudio_tracked_hip('re')        ~  acquisition/claudio_aligned_code_re_separate.m
%   claudio_tracked_hip('lo')        ~  acquisition/claudio_aligned_code_lo_separate.m
%
% This file holds the JOB CONTRACT only — which captures belong to the job, which code file, what the result file is
% called and which variables it holds, when the capture is archived.  Everything between fopen and save in the
% reference scripts (carrier search, per-chunk carrier, the 40-ms code loop with re-alignment) runs inside
% libtwstft_hip.so through ONE call per capture, twstft_tracked_mex (mex/twstft_tracked_mex.cpp -> twx_tracked_file).
%
% Environment, as in the reference scripts: OP (station flag, 0/1), processing_dir (captures), codelocation (n*.bin code
% files), remotechannel (1/2: suffix of the capture files of the remote receiver channel; the local job reads the other).
% Result: <prefix><capture>.mat in the current directory with the reference's variable set
%   df indice1 correction1 SNR1r SNR1i puissance1 puissancecode puissancenoise code xval1 moved movedval
% An existing result (.mat or .mat.gz) means the capture is skipped.  Shell use: octave -q --eval "claudio_tracked_hip('lo')"
function claudio_tracked_hip(flavour)
  if (nargin < 1) flavour = 'ranging'; end
  site = site_settings();
  job = job_rules(flavour, site);
  todo = dir(fullfile(site.captures, sprintf('*_%d.bin', job.channel)));
  codefiles = dir(fullfile(site.codes, 'n*.bin'));
  if (numel(codefiles) < 2) error('claudio_tracked_hip: need the two station codes n*.bin in %s', site.codes); end
  chipfile = fullfile(site.codes, codefiles(job.codeslot).name);
  codeb = slurp_bytes(chipfile);
  code = 2 * reshape([codeb(:)'; codeb(:)'], 1, []) - 1;      % the +-1 replica at 2 samples per chip, stored with the results
  for n = 1:numel(todo)
    capture = todo(n).name;
    stem = strrep(capture, '.bin', '.mat');
    result = [job.prefix, stem];
    if (exist(result, 'file') || exist([result, '.gz'], 'file'))
      printf('%s already done\n', result);
      continue
    end
    printf('%s -> %s (code %s)\n', capture, result, codefiles(job.codeslot).name);
    [xval1, indice1, correction1, SNR1r, SNR1i, puissance1, df, moved, movedval, kbon, puissancecode, puissancenoise] = ...
        twstft_tracked_mex(fullfile(site.captures, capture), codeb, flavour, site.OP);
    if (job.needs_carrier && kbon == 0)
      printf('%s: no carrier found in the search band, no codes measured\n', capture);
    end
    report(indice1, correction1, puissance1, SNR1r, SNR1i, df, numel(moved));
    save('-mat', result, 'df', 'indice1', 'correction1', 'SNR1r', 'SNR1i', 'puissance1', 'puissancecode', ...
         'puissancenoise', 'code', 'xval1', 'moved', 'movedval');
    archive_capture(site, job, capture, stem);
  end
end

% ---- the site: environment variables with the reference's fall-backs ------------------------------------------
function site = site_settings()
  site.OP = env_number('OP', 0);
  site.remotechannel = env_number('remotechannel', 2);
  site.captures = env_text('processing_dir', './');
  site.codes = env_text('codelocation', './codes/');
end
function v = env_number(name, fallback)
  t = getenv(name);
  if (isempty(t)) v = fallback; else v = str2double(t); end
end
function t = env_text(name, fallback)
  t = getenv(name);
  if (isempty(t)) t = fallback; end
end

% ---- what tells the three jobs apart ----------------------------------------------------------------------------
%   channel   : capture suffix (remote receiver channel for ranging/re, the other one for lo)
%   codeslot  : position in dir('n*.bin') — the station's own code or the partner's (parity rule of the reference)
%   prefix    : result file prefix
%   archive   : 'pair' = move the capture to donetw/ once the sibling job's result exists, 'always' = right away
function job = job_rules(flavour, site)
  switch (flavour)
    case 'ranging'
      job = struct('channel', site.remotechannel, 'codeslot', mod(site.OP, 2) + 1, 'prefix', 'rangingclaudio', ...
                   'archive', 'pair', 'sibling', '*remote*', 'needs_carrier', true);
    case 're'
      job = struct('channel', site.remotechannel, 'codeslot', mod(site.OP + 1, 2) + 1, 'prefix', 'remoteclaudio', ...
                   'archive', 'pair', 'sibling', '*ranging*', 'needs_carrier', true);
    case 'lo'
      job = struct('channel', 3 - site.remotechannel, 'codeslot', mod(site.OP, 2) + 1, 'prefix', 'localclaudio', ...
                   'archive', 'always', 'sibling', '', 'needs_carrier', false);
    otherwise
      error('claudio_tracked_hip: flavour must be ranging, re or lo');
  end
end

function b = slurp_bytes(path)
  h = fopen(path, 'r');
  if (h < 0) error('claudio_tracked_hip: cannot open %s', path); end
  b = fread(h, inf, 'uint8=>uint8');
  fclose(h);
end

% one summary line per capture (the reference prints one line per code; the per-code values are in the .mat)
function report(indice1, correction1, puissance1, SNR1r, SNR1i, df, nmoved)
  if (isempty(indice1)) return; end
  snr_db = 10 * log10(SNR1r + SNR1i);
  printf('  %d codes, %d carrier updates (%.3f .. %.3f Hz), %d re-alignments, median SNR %.1f dB, mean power %.1f dB\n', ...
         numel(indice1), numel(df), min(df), max(df), nmoved, median(snr_db), 10 * log10(mean(puissance1)));
end

function archive_capture(site, job, capture, stem)
  ready = strcmp(job.archive, 'always');
  if (!ready) ready = !isempty(dir([job.sibling, stem, '*'])); end
  if (ready)
    [ok, msg] = movefile(fullfile(site.captures, capture), fullfile(site.captures, 'donetw'));
    if (!ok) printf('  could not archive %s: %s\n', capture, msg); end
  end
end
