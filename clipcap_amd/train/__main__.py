from clipcap_amd.train.train import start_training

if __name__ == "__main__":
    raise SystemExit(start_training())
