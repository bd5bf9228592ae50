% claudio_aligned_code_ranging_separate_hip.m — drop-in for acquisition/claudio_aligned_code_ranging_separate.m
% with processing(d,df) (the per-code correlation, :49-102) on the GPU through twstft_processing_mex.
%
% Flow and bookkeeping follow the reference script (:104-220): captures *_<remotechannel>.bin (int16 [I Q]), code
% n*.bin picked by the parity of OP+remote+2*ranging, 30 s skipped, 2-s chunks; search_df on the first chunk(s), then
% per chunk the carrier from the 7 bins around kbon and the tracked loop over 40-ms codes with re-alignment; results
% saved to (ranging|remote|local)claudio<capture>.mat with the reference's variable list.  search_df (:27-47) and the
% long fft(d.^2) stay in Octave: they need the whole correlation map / a 10^7-point spectrum once per lock, not the
% per-code hot loop.  (The all-GPU version of the whole flow, including search_df, is amaranth_twstft_amd/tracked.py.)
1;
pkg load signal
format long
global temps freq fcode code fs Nint codeb
fs=5e6;
Nint=1;
remote=0;
ranging=1;
ls=2;
df_threshold=20;
OP=getenv('OP');
datalocation=getenv('processing_dir');
codelocation=getenv('codelocation');
remotechannel=getenv('remotechannel');
if (isempty(codelocation)) codelocation='./codes/'; end
if (isempty(OP)) OP=0; else OP=str2num(OP); end
if (isempty(datalocation)) datalocation='./'; end
if (isempty(remotechannel)) remotechannel=2; else remotechannel=str2num(remotechannel); end

function k=search_df(d,k,df_threshold)               % claudio…separate.m:27-47, unchanged arithmetic
  global freq fcode temps
  kbon=0;
  d2=fftshift(abs(fft(d.^2)));
  cand=find(d2(k)>median(d2(k))*df_threshold)+k(1)-1;
  if (length(cand)>0 && length(cand)<100)
    for c=1:length(cand)
      lo=exp(-j*2*pi*(freq(cand(c))/2)*temps);
      prnmap=abs(ifft(fcode.*conj(fft(d(1:length(fcode)).*lo))));
      [prnsig,b]=max(prnmap);
      prnmap(b-5:b+5)=0;
      if ((prnsig^2/var(prnmap))>100) kbon=cand(c); end
    end
  end
  k=kbon;
end

% processing(d,df) of claudio…separate.m:49 — d: one code length, mean removed by the caller
function [xval,indice,correction,SNRr,SNRi,puissance,puissancecode,puissancenoise]=processing(d,df)
  global fs Nint codeb
  [xval,indice,correction,SNRr,SNRi,puissance,puissancecode,puissancenoise]=twstft_processing_mex(d,df,codeb,fs,Nint,'claudio');
end

captures=dir([datalocation,'/*_',num2str(remotechannel),'.bin']);
codes=dir([codelocation,'/n*.bin']);
for c=1:length(captures)
  codename=codes(mod(OP+remote+ranging*2,2)+1).name;  % LTFB=odd OP=even
  base=strrep(captures(c).name,'.bin','.mat');
  if (remote==1) nomout=['remoteclaudio',base]; elseif (ranging==1) nomout=['rangingclaudio',base]; else nomout=['localclaudio',base]; end
  if ((exist(nomout)!=0) || (exist([nomout,'.gz'])!=0))
    printf("%s already done\n",nomout);
    continue
  end
  fc=fopen([codelocation,'/',codename]);
  codeb=fread(fc,inf,'uint8');
  fclose(fc);
  code=2*repelems(codeb,[[1:length(codeb)];2*ones(1,length(codeb))])-1;
  fcode=fft(code');                                  % search_df only
  n=length(code);
  printf("%s\n",captures(c).name);
  f=fopen([datalocation,'/',captures(c).name]);
  fseek(f,30*fs*2*2);                                % skip 30 s (:128)
  temps=[0:n-1]'/fs;
  freq=linspace(-fs/2,fs/2-fs/fs,fs*ls);
  printf("n\tdt1\tdf1\tP1\tSNR1\tdt2\tdf2\tP2\tSNR2\r\n");
  if (ranging==1)
    k=find((freq<8000)&(freq>-8000));
  elseif (OP==1)
    k=find((freq>-108000)&(freq<-92000));
  else
    k=find((freq<108000)&(freq>92000));
  end
  clear xval1 indice1 correction1 SNR1r SNR1i puissance1 df dindex
  dold=[]; moved=[]; movedval=[]; df_found=0; p=1; pfreq=1;
  do
    d=fread(f,fs*2*ls,'int16');
    longueur=length(d);
    if (longueur==fs*2*ls)
      d=d(1:2:end)+j*d(2:2:end);
      if (df_found==0)
        kbon=search_df(d,k,df_threshold);
        if (kbon!=0) df_found=1; end
        fclose(f);                                   % the reference re-reads the file from its start here (:153-155)
        f=fopen([datalocation,'/',captures(c).name]);
        d=fread(f,fs*2*ls,'int16');
        d=d(1:2:end)+j*d(2:2:end);
      end
      if (df_found==1)
        d=[dold ; d];
        d2=fftshift(abs(fft(d.^2)));
        [~,m]=max(d2(kbon-3:kbon+3));
        df(pfreq)=freq(m+kbon-3-1)/2;
        dindex=1;
        do
          dpart=d(round(dindex):round(dindex)+n-1); dpart=dpart-mean(dpart);
          [xval1(p),indice1(p),correction1(p),SNR1r(p),SNR1i(p),puissance1(p),puissancecode,puissancenoise]=processing(dpart,df(pfreq));
          indice1(p)=indice1(p)/(2*Nint+1);
          if (10*log10(SNR1i(p)+SNR1r(p))>-30)
            if (((indice1(p)>43)&&(indice1(p)<n/2)) || ((indice1(p)<n-2)&&(indice1(p)>n/2)))
              printf("MOVED %d\n",indice1(p));
              moved=[moved p];
              movedval=[movedval indice1(p)+1];
              if ((dindex-indice1(p)+1)<0) dindex=dindex+n; end
              dindex=dindex-indice1(p)+21;
              dpart=d(round(dindex):round(dindex)+n-1); dpart=dpart-mean(dpart);
              [xval1(p),indice1(p),correction1(p),SNR1r(p),SNR1i(p),puissance1(p),puissancecode,puissancenoise]=processing(dpart,df(pfreq));
            end
          end
          printf("%d\t%.12f\t%.3f\t%.1f\t%.1f\r\n",p,(indice1(p)-1+correction1(p))/fs/(2*Nint+1),df(pfreq),10*log10(puissance1(p)),10*log10(SNR1i(p)+SNR1r(p)));
          p=p+1;
          dindex=dindex+n;
        until (dindex+n-1>length(d))
      end
    end
    if (exist('dindex'))
      if (dindex<length(d)) dold=d(round(dindex):end); else dold=[]; end
      pfreq=pfreq+1;
    end
  until (longueur!=fs*2*ls)
  fclose(f);
  eval(['save -mat ',nomout,' corr* df indic* SNR* code puissan* xval* moved*']);
end
