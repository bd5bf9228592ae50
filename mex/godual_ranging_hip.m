% godual_ranging_hip.m — drop-in for processing/Octave/godual_ranging.m with the correlation on the GPU.
%
% Same contract as the reference script (godual_ranging.m:57-133): every capture 1*.bin of `datalocation` (int16
% [I1 Q1 I2 Q2]) is correlated, one code length at a time, against the code n*.bin picked by the parity of OP+remote;
% one TSV row per window on stdout, the quadratic-fit residual statistics, and <capture>.mat (remote<capture>.mat when
% remote=1) with the variables corr* df1 df2 indic* SNR* code puissan* xval*.  processing(d,k) keeps its signature
% (:12) and is what runs on the GPU, through twstft_processing_mex (build: see INTEGRATION.md).
%
% What differs from the reference: windows are handed to the GPU WIN at a time (the outputs of the MEX call are
% vectors), both channels from one upload; an already existing .mat is skipped (as the newer scripts do).
% Environment variables OP, processing_dir, codelocation are honoured like in acquisition/*.m.
1;
pkg load signal
global fs Nint codeb
fs=5e6;
Nint=1;
remote=0;
OP=0;
WIN=16;                      % windows per GPU call
datalocation=getenv('processing_dir');
codelocation=getenv('codelocation');
if (isempty(datalocation)) datalocation='./'; end
if (isempty(codelocation)) codelocation='./codes/'; end
if (!isempty(getenv('OP'))) OP=str2num(getenv('OP')); end
ngpu=1;                      % TWX_NGPU=N: every capture as ONE call over N GPUs (call form C of the gateway: each device reads its own extent)
if (!isempty(getenv('TWX_NGPU'))) ngpu=str2num(getenv('TWX_NGPU')); end

% processing(d,k) of godual_ranging.m:12 — d: complex column, mean removed by the caller, one or several code lengths
function [indice,correction,SNRr,SNRi,df,puissance,puissancecode,puissancenoise,xval]=processing(d,k)
  global fs Nint codeb
  [indice,correction,SNRr,SNRi,df,puissance,puissancecode,puissancenoise,xval]=twstft_processing_mex(d,[k(1) k(end)],codeb,fs,Nint);
end

captures=dir([datalocation,'/1*.bin']);
codes=dir([codelocation,'/n*.bin']);
for c=1:length(captures)
  codename=codes(mod(OP+remote,2)+1).name;          % LTFB=odd OP=even (godual_ranging.m:60)
  fc=fopen([codelocation,'/',codename]);
  codeb=fread(fc,inf,'uint8');                      % chips 0/1, before repelems
  fclose(fc);
  code=2*repelems(codeb,[[1:length(codeb)];2*ones(1,length(codeb))])-1;   % saved with the results, as the reference does
  n=length(code);
  base=strrep(captures(c).name,'.bin','.mat');
  if (remote==1) matname=[datalocation,'/remote',base]; else matname=[datalocation,'/',base]; end
  if (exist(matname,'file') || exist([matname,'.gz'],'file'))
    printf("%s already done\n",matname);
    continue
  end
  printf("%s\n",captures(c).name);
  freq=linspace(-fs/2,fs/2,n);                      % godual_ranging.m:73
  if (remote!=1)
    k=find((freq<20000)&(freq>-20000));
  elseif (OP==1)
    k=find((freq>-120000)&(freq<-80000));
  else
    k=find((freq<120000)&(freq>80000));
  end
  clear indice1 correction1 SNR1r SNR1i df1 puissance1 puissance1code puissance1noise xval1
  clear indice2 correction2 SNR2r SNR2i df2 puissance2 puissance2code puissance2noise xval2
  f=fopen([datalocation,'/',captures(c).name]);
  printf("n\tdt1\tdf1\tP1\tSNR1\tdt2\tdf2\tP2\tSNR2\r\n");
  p=0;
  do
    if (ngpu>1)
      % the whole capture in one call: the library cuts it into ngpu contiguous extents, one per device, and gathers the records
      chan=0; if (remote==1) chan=1; end
      [ii,cc,sr,si,dd,pu,pc,pn,xv]=twstft_processing_mex('file',[datalocation,'/',captures(c).name],2,chan,[k(1) k(end)],codeb,fs,Nint,ngpu);
      nw=columns(ii); got=0;
    else
      [raw,got]=fread(f,n*4*WIN,'int16=>int16');
      nw=floor(got/(n*4));
    end
    if (nw>0)
      if (ngpu>1)
        % outputs already there
      elseif (remote!=1)
        raw=raw(1:nw*n*4);
        % rows of every output: channel 1 (measurement), channel 2 (reference); columns: windows
        [ii,cc,sr,si,dd,pu,pc,pn,xv]=twstft_processing_mex(raw,2,0,[k(1) k(end)],codeb,fs,Nint);
      else
        raw=raw(1:nw*n*4);
        [ii,cc,sr,si,dd,pu,pc,pn,xv]=twstft_processing_mex(raw,2,1,[k(1) k(end)],codeb,fs,Nint);
      end
      for w=1:nw
        p=p+1;
        indice1(p)=ii(1,w); correction1(p)=cc(1,w); SNR1r(p)=sr(1,w); SNR1i(p)=si(1,w); df1(p)=dd(1,w);
        puissance1(p)=pu(1,w); puissance1code(p)=pc(1,w); puissance1noise(p)=pn(1,w); xval1(p)=xv(1,w);
        if (remote!=1)
          indice2(p)=ii(2,w); correction2(p)=cc(2,w); SNR2r(p)=sr(2,w); SNR2i(p)=si(2,w); df2(p)=dd(2,w);
          puissance2(p)=pu(2,w); puissance2code(p)=pc(2,w); puissance2noise(p)=pn(2,w); xval2(p)=xv(2,w);
          printf("%d\t%.12f\t%.3f\t%.1f\t%.1f\t%.12f\t%.3f\t%.1f\t%.1f\r\n",p,(indice1(p)-1+correction1(p))/fs/(2*Nint+1),df1(p),10*log10(puissance1(p)),10*log10(SNR1i(p)+SNR1r(p)),(indice2(p)-1-correction2(p))/fs/(2*Nint+1),df2(p),10*log10(puissance2(p)),10*log10(SNR2i(p)+SNR2r(p)));
        else
          printf("%d\t%.12f\t%.3f\t%.1f\t%.1f\r\n",p,(indice1(p)-1+correction1(p))/fs/(2*Nint+1),df1(p),10*log10(puissance1(p)),10*log10(SNR1i(p)+SNR1r(p)));
        end
      end
    end
  until (got<n*4*WIN)
  fclose(f);
  if (p>3)                                           % godual_ranging.m:104-113
    if (remote!=1)
      s2=(indice2-1+correction2)/(2*Nint+1)/fs;
      [a,b]=polyfit([1:length(s2)],s2,2);            % should be flat: loop-back channel
      std(s2-b.yf)
      mean(s2-b.yf)
    end
    s1=(indice1-1+correction1)/(2*Nint+1)/fs;
    [a,b]=polyfit([1:length(s1)],s1,2);
    std(s1-b.yf)
    mean(s1-b.yf)
  end
  if (p>0)
    eval(['save -mat ',matname,' corr* df1 df2 indic* SNR* code puissan* xval*']);
  end
end
